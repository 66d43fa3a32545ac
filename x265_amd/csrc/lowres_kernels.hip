// Half-resolution downscale with the H / V / HV half-pel planes and their
// replicated borders, in one pass over the PADDED output domain.
//
// Replaces primitives.frameInitLowres + 4x extendPicBorder as called from
// Lowres::init (source/common/lowres.cpp:380-386; arithmetic
// source/common/pixel.cpp:596-620, borders source/common/pixel.cpp:1044-1058).
// A border sample equals the nearest picture sample, so every output position
// is computed straight from the source at its clamped picture coordinate: no
// second pass, no dependency between threads, fully coalesced row stores.
// HBM-bound: algorithmic traffic is one read of the source plus one write of
// four quarter-size planes.
#include "device_common.h"
#include "kernels.h"

namespace xg {

__device__ __forceinline__ int ravg(int a, int b) { return (a + b + 1) >> 1; }

template <typename P>
struct DownscaleArgs
{
    const P* src;
    long long srcStride;
    int srcWidth, srcHeight;   // valid source extent; beyond it the edge sample repeats
    P* dst0;                   // origin (pixel 0,0) of output plane 0
    long long planeSize;
    int dstStride;
    int width, lines;          // output picture size
    int marginX, marginY;
};

// The two 2x2 filters: the lookahead's cascaded rounding averages
// (source/common/pixel.cpp:607) and the plain rounded mean the temporal-filter
// path uses instead (source/common/pixel.cpp:1026).  `v` is the vertical
// partial of a column (rounded average, or plain sum for the mean).
template <bool MEAN> __device__ __forceinline__ int vpart(int a, int b) { return MEAN ? a + b : ravg(a, b); }
template <bool MEAN> __device__ __forceinline__ int hpart(int a, int b) { return MEAN ? (a + b + 2) >> 2 : ravg(a, b); }

// ---- packed arithmetic for the interior of the lookahead filter ------------------------------
// (a + b + 1) >> 1 on every pixel of a dword: v_lerp_u8 with a rounding bit per byte for 8-bit pixels,
// packed 16-bit adds for deeper ones (two 12-bit samples and the rounding bit stay inside 16 bits)
template <typename P> __device__ __forceinline__ uint32_t pavg(uint32_t a, uint32_t b);
template <> __device__ __forceinline__ uint32_t pavg<uint8_t>(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_lerp(a, b, 0x01010101u);
}
template <> __device__ __forceinline__ uint32_t pavg<uint16_t>(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 one = { 1, 1 };
    const u16x2 r = (__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b) + one) >> one;
    return __builtin_bit_cast(uint32_t, r);
}
// the even / the odd pixels of two consecutive dwords, as one dword
template <typename P> __device__ __forceinline__ uint32_t evens(uint32_t lo, uint32_t hi)
{
    return __builtin_amdgcn_perm(hi, lo, sizeof(P) == 1 ? 0x06040200u : 0x05040100u);
}
template <typename P> __device__ __forceinline__ uint32_t odds(uint32_t lo, uint32_t hi)
{
    return __builtin_amdgcn_perm(hi, lo, sizeof(P) == 1 ? 0x07050301u : 0x07060302u);
}
// dword k of a row moved left by two pixels
template <typename P> __device__ __forceinline__ uint32_t shift2(const uint32_t* d, int k)
{
    return sizeof(P) == 1 ? __builtin_amdgcn_alignbyte(d[k + 1], d[k], 2) : d[k + 1];
}

// One output row of two planes from the vertical partials `v` of 16 (+1) source columns:
// plane A = avg(even, odd), plane B = avg(odd, next even)  (source/common/pixel.cpp:607-616)
template <typename P> __device__ __forceinline__ void hpair(const uint32_t* v, Row<P>& a, Row<P>& b)
{
    constexpr int NW = PixT<P>::NW;
#pragma unroll
    for (int k = 0; k < NW; k++)
    {
        const uint32_t e = evens<P>(v[2 * k], v[2 * k + 1]), o = odds<P>(v[2 * k], v[2 * k + 1]);
        const uint32_t e1 = evens<P>(shift2<P>(v, 2 * k), shift2<P>(v, 2 * k + 1));
        a.w[k] = pavg<P>(e, o);
        b.w[k] = pavg<P>(o, e1);
    }
}

// One thread: 8 consecutive output positions of one padded row, all 4 planes.
template <typename P, bool MEAN>
__device__ void downscale_span(const DownscaleArgs<P>& a, int span, int prow)
{
    const int px0 = span * 8 - a.marginX;      // first output x of the span (may be negative)
    const int py = prow - a.marginY;
    const int y = clampi(py, 0, a.lines - 1);
    const int r0 = min(2 * y, a.srcHeight - 1), r1 = min(2 * y + 1, a.srcHeight - 1),
              r2 = min(2 * y + 2, a.srcHeight - 1);
    const P* s0 = a.src + (long long)r0 * a.srcStride;
    const P* s1 = a.src + (long long)r1 * a.srcStride;
    const P* s2 = a.src + (long long)r2 * a.srcStride;

    P* d = a.dst0 + (long long)py * a.dstStride + px0;
    int o0[8], oh[8], ov[8], oc[8];
    const bool interior = px0 >= 0 && px0 + 8 <= a.width;
    // A span inside the left / right border repeats one pixel of the first / last picture span (margins and
    // widths are multiples of 8, so a span never straddles); the stride-rounding slack is left to the slow path.
    const bool left = px0 + 8 <= 0, right = px0 >= a.width && px0 + 8 <= a.width + a.marginX;
    const int pxc = left ? 0 : right ? a.width - 8 : px0;
    if (!MEAN && (interior || left || right) && a.width >= 8 && 2 * pxc + 15 < a.srcWidth)
    {
        // the bulk of the picture, on packed pixels: 16 source columns of three rows plus the one column
        // behind them (the last column of the picture repeats, as the reference's source padding has it),
        // two vertical averages per dword, then the even / odd split for the four planes
        constexpr int NW = PixT<P>::NW;
        uint32_t t0[2 * NW + 1], t1[2 * NW + 1], t2[2 * NW + 1];
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const Row<P> r0 = load_row<P>(s0 + 2 * pxc + 8 * k), r1 = load_row<P>(s1 + 2 * pxc + 8 * k),
                         r2 = load_row<P>(s2 + 2 * pxc + 8 * k);
#pragma unroll
            for (int j = 0; j < NW; j++)
            {
                t0[k * NW + j] = r0.w[j];
                t1[k * NW + j] = r1.w[j];
                t2[k * NW + j] = r2.w[j];
            }
        }
        const int last = min(2 * pxc + 16, a.srcWidth - 1);
        t0[2 * NW] = s0[last];
        t1[2 * NW] = s1[last];
        t2[2 * NW] = s2[last];
        uint32_t v01[2 * NW + 1], v12[2 * NW + 1];
#pragma unroll
        for (int j = 0; j <= 2 * NW; j++)
        {
            v01[j] = pavg<P>(t0[j], t1[j]);
            v12[j] = pavg<P>(t1[j], t2[j]);
        }
        Row<P> p0, ph, pv, pc;
        hpair<P>(v01, p0, ph);
        hpair<P>(v12, pv, pc);
        if (left | right)
        {
            Row<P>* planes[4] = { &p0, &ph, &pv, &pc };
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                constexpr int TOP = PixT<P>::NW - 1, BITS = 8 * (int)sizeof(P);
                const uint32_t px = left ? planes[q]->w[0] & ((1u << BITS) - 1) : planes[q]->w[TOP] >> (32 - BITS);
                const uint32_t rep = sizeof(P) == 1 ? px * 0x01010101u : px * 0x00010001u;
#pragma unroll
                for (int j = 0; j <= TOP; j++)
                    planes[q]->w[j] = rep;
            }
        }
        store_row<P>(d, p0);
        store_row<P>(d + a.planeSize, ph);
        store_row<P>(d + 2 * a.planeSize, pv);
        store_row<P>(d + 3 * a.planeSize, pc);
        return;
    }
    if (interior && 2 * px0 + 23 < a.srcWidth)
    {
        // 18 source columns per row, fetched as three 8-pixel rows
        int t0[24], t1[24], t2[24];
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
            unpack_row<P>(load_row<P>(s0 + 2 * px0 + 8 * k), t0 + 8 * k);
            unpack_row<P>(load_row<P>(s1 + 2 * px0 + 8 * k), t1 + 8 * k);
            unpack_row<P>(load_row<P>(s2 + 2 * px0 + 8 * k), t2 + 8 * k);
        }
        int v01[18], v12[18];
#pragma unroll
        for (int k = 0; k < 18; k++)
        {
            v01[k] = vpart<MEAN>(t0[k], t1[k]);
            v12[k] = vpart<MEAN>(t1[k], t2[k]);
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            o0[i] = hpart<MEAN>(v01[2 * i], v01[2 * i + 1]);
            oh[i] = hpart<MEAN>(v01[2 * i + 1], v01[2 * i + 2]);
            ov[i] = hpart<MEAN>(v12[2 * i], v12[2 * i + 1]);
            oc[i] = hpart<MEAN>(v12[2 * i + 1], v12[2 * i + 2]);
        }
    }
    else
    {
        const int slackStart = a.width + a.marginX;   // first x right of the right margin
        int prevx = -1;
        int l0 = 0, lh = 0, lv = 0, lc = 0;   // values at prevx
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const int px = px0 + i;
            const int x = clampi(px, 0, a.width - 1);
            if (x != prevx)
            {
                const int c0 = min(2 * x, a.srcWidth - 1), c1 = min(2 * x + 1, a.srcWidth - 1),
                          c2 = min(2 * x + 2, a.srcWidth - 1);
                const int a0 = vpart<MEAN>(s0[c0], s1[c0]), a1 = vpart<MEAN>(s0[c1], s1[c1]), a2 = vpart<MEAN>(s0[c2], s1[c2]);
                const int b0 = vpart<MEAN>(s1[c0], s2[c0]), b1 = vpart<MEAN>(s1[c1], s2[c1]), b2 = vpart<MEAN>(s1[c2], s2[c2]);
                l0 = hpart<MEAN>(a0, a1);
                lh = hpart<MEAN>(a1, a2);
                lv = hpart<MEAN>(b0, b1);
                lc = hpart<MEAN>(b1, b2);
                prevx = x;
            }
            // stride-rounding slack right of the margin stays zero, as calloc leaves it
            const bool slack = px >= slackStart;
            o0[i] = slack ? 0 : l0;
            oh[i] = slack ? 0 : lh;
            ov[i] = slack ? 0 : lv;
            oc[i] = slack ? 0 : lc;
        }
    }

    store_row<P>(d, pack_row<P>(o0));
    store_row<P>(d + a.planeSize, pack_row<P>(oh));
    store_row<P>(d + 2 * a.planeSize, pack_row<P>(ov));
    store_row<P>(d + 3 * a.planeSize, pack_row<P>(oc));
}

// grid: (ceil(spans*rows / 256), n jobs).  `level` 0: full-res -> half-res,
// 1: half-res plane 0 -> quarter-res (frameInitLowerRes, lowres.cpp:393-398).
template <typename P>
__global__ void __launch_bounds__(256) lowres_init_kernel(const LowresJob* jobs, Geometry g, int level)
{
    const LowresJob job = jobs[blockIdx.y];
    DownscaleArgs<P> a;
    if (level == 0)
    {
        a.src = (const P*)job.src;
        a.srcStride = job.srcStride;
        a.srcWidth = g.srcWidth;
        a.srcHeight = g.srcHeight;
        a.dst0 = (P*)job.planes + g.padOffset;
        a.planeSize = g.planeSize;
        a.dstStride = g.stride;
        a.width = g.width;
        a.lines = g.lines;
        a.marginX = g.marginX;
        a.marginY = g.marginY;
    }
    else
    {
        // the half-res plane is read through its own replicated border, which
        // equals clamping at its picture edge
        a.src = (const P*)job.planes + g.padOffset;
        a.srcStride = g.stride;
        a.srcWidth = g.width;
        a.srcHeight = g.lines;
        a.dst0 = (P*)job.planes4 + g.padOffset4;
        a.planeSize = g.planeSize4;
        a.dstStride = g.stride4;
        a.width = g.width4;
        a.lines = g.lines4;
        a.marginX = g.marginX4;
        a.marginY = g.marginY4;
    }
    const int spans = a.dstStride / 8;
    const int rows = a.lines + 2 * a.marginY;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= spans * rows)
        return;
    if (g.mcstfFilter)
        downscale_span<P, true>(a, idx % spans, idx / spans);
    else
        downscale_span<P, false>(a, idx % spans, idx / spans);
}

hipError_t launch_lowres_init(hipStream_t s, const Geometry& g, int bpp, const LowresJob* dJobs, int n)
{
    if (n <= 0)
        return hipSuccess;
    {
        const int work = (g.stride / 8) * (g.lines + 2 * g.marginY);
        dim3 grid((work + 255) / 256, n);
        if (bpp == 1)
            hipLaunchKernelGGL(lowres_init_kernel<uint8_t>, grid, dim3(256), 0, s, dJobs, g, 0);
        else
            hipLaunchKernelGGL(lowres_init_kernel<uint16_t>, grid, dim3(256), 0, s, dJobs, g, 0);
    }
    if (g.stride4 > 0)
    {
        const int work = (g.stride4 / 8) * (g.lines4 + 2 * g.marginY4);
        dim3 grid((work + 255) / 256, n);
        if (bpp == 1)
            hipLaunchKernelGGL(lowres_init_kernel<uint8_t>, grid, dim3(256), 0, s, dJobs, g, 1);
        else
            hipLaunchKernelGGL(lowres_init_kernel<uint16_t>, grid, dim3(256), 0, s, dJobs, g, 1);
    }
    return hipGetLastError();
}

} // namespace xg
