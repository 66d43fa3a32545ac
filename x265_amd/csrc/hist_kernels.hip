// Picture statistics for histogram-based scene-cut detection.
//
// Replaces the pixel passes of LookaheadTLD::collectPictureStatistics
// (source/encoder/slicetype.cpp:1550-1834) for a 4:2:0 picture: 256-bin
// histograms and sample sums of the 4x4 picture segments - luma on the
// quarter-sample picture that Lowres::init derives from the half-res plane
// (source/common/lowres.cpp:40-57, 404-414), chroma on every fourth sample of
// every fourth row (calculateHistogram with dsFactor 4, :1655-1683) - and the
// per-block-row sums of 8x8 luma / 4x4 chroma block variances
// (computePictureStatistics, :1567-1650).  The quarter-sample picture is only
// ever read by the histogram, so it is computed on the fly and never stored.
// The reference's rounding and averaging of these counts stays on the host.
#include "device_common.h"
#include "kernels.h"

namespace xg {

__device__ __forceinline__ int ravg2(int a, int b) { return (a + b + 1) >> 1; }

// grid (row chunks, 16 segments, 3 planes)
template <typename P>
__global__ void __launch_bounds__(256) hist_kernel(HistJob job)
{
    __shared__ uint32_t bins[256];
    __shared__ unsigned long long total;
    const int seg = blockIdx.y, sx = seg >> 2, sy = seg & 3, plane = blockIdx.z;
    bins[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        total = 0;
    __syncthreads();

    const int shift = job.depth - 8;
    unsigned long long mine = 0;
    if (plane == 0)
    {
        const int qw = job.width / 4, qh = job.height / 4;
        const int segW = qw / 4, segH = qh / 4;
        const int cols = segW + (sx == 3 ? qw - 4 * segW : 0), rows = segH + (sy == 3 ? qh - 4 * segH : 0);
        const P* src = (const P*)job.lowres0;
        for (int r = blockIdx.x; r < rows; r += gridDim.x)
        {
            const P* r0 = src + (long long)(2 * (sy * segH + r)) * job.lowresStride + 2 * (sx * segW);
            const P* r1 = r0 + job.lowresStride;
            for (int c = threadIdx.x; c < cols; c += 256)
            {
                const int q = ravg2(ravg2(r0[2 * c], r1[2 * c]), ravg2(r0[2 * c + 1], r1[2 * c + 1]));
                const int val = q >> shift;
                atomicAdd(&bins[val], 1u);
                mine += (unsigned)val;
            }
        }
    }
    else
    {
        const int segW = job.width / 4, segH = job.height / 4;
        const int w = (segW + (sx == 3 ? job.width - 4 * segW : 0)) >> 1, h = (segH + (sy == 3 ? job.height - 4 * segH : 0)) >> 1;
        const int cols = (w + 3) >> 2, rows = (h + 3) >> 2;
        const P* src = (const P*)(plane == 1 ? job.u : job.v) + (long long)((sy * segH) >> 1) * job.strideC + ((sx * segW) >> 1);
        for (int r = blockIdx.x; r < rows; r += gridDim.x)
        {
            const P* row = src + (long long)(4 * r) * job.strideC;
            for (int c = threadIdx.x; c < cols; c += 256)
            {
                const int val = row[4 * c] >> shift;
                atomicAdd(&bins[val], 1u);
                mine += (unsigned)val;
            }
        }
    }
    if (mine)
        atomicAdd(&total, mine);
    __syncthreads();
    const int cell = (seg * 3 + plane);
    if (bins[threadIdx.x])
        atomicAdd(job.hist + cell * 256 + threadIdx.x, bins[threadIdx.x]);
    if (threadIdx.x == 0 && total)
        atomicAdd(job.sums + cell, total);
}

// One thread per block (8x8 luma, 4x4 chroma); grid (blocks in a row / 256, block rows, 3 planes).
// Samples beyond the picture are the replicated edge, as PicYuv::copyFromPicture pads them.
template <typename P>
__global__ void __launch_bounds__(256) pic_variance_kernel(HistJob job)
{
    const int plane = blockIdx.z;
    const int size = plane ? 4 : 8, shift = plane ? 4 : 6;
    const int w = plane ? job.width >> 1 : job.width, h = plane ? job.height >> 1 : job.height;
    const int blocksX = (w + size - 1) / size, blocksY = (h + size - 1) / size;
    const int by = blockIdx.y, bx = blockIdx.x * 256 + threadIdx.x;
    unsigned long long var = 0;
    if (by < blocksY && bx < blocksX)
    {
        const P* src = (const P*)(plane == 0 ? job.y : (plane == 1 ? job.u : job.v));
        const long long stride = plane ? job.strideC : job.strideY;
        uint32_t sum = 0, ssd = 0;
        for (int y = 0; y < size; y++)
        {
            const P* row = src + (long long)min(by * size + y, h - 1) * stride;
            for (int x = 0; x < size; x++)
            {
                const uint32_t v = row[min(bx * size + x, w - 1)];
                sum += v;
                ssd += v * v;
            }
        }
        var = ssd - (uint32_t)(((unsigned long long)sum * sum) >> shift);      // acEnergyVarHist, slicetype.cpp:92-98
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        var += __shfl_down(var, off);
    if ((threadIdx.x & 63) == 0 && var && by < blocksY)
        atomicAdd(job.rowVar + plane * job.maxRows + by, var);
}

hipError_t launch_hist_stats(hipStream_t s, int bpp, const HistJob& job)
{
    {
        dim3 grid(32, 16, 3);
        if (bpp == 1)
            hipLaunchKernelGGL(hist_kernel<uint8_t>, grid, dim3(256), 0, s, job);
        else
            hipLaunchKernelGGL(hist_kernel<uint16_t>, grid, dim3(256), 0, s, job);
    }
    {
        const int blocksX = (job.width + 7) / 8, blocksY = (job.height + 7) / 8;   // the chroma grids are no larger
        dim3 grid((blocksX + 255) / 256, blocksY, 3);
        if (bpp == 1)
            hipLaunchKernelGGL(pic_variance_kernel<uint8_t>, grid, dim3(256), 0, s, job);
        else
            hipLaunchKernelGGL(pic_variance_kernel<uint16_t>, grid, dim3(256), 0, s, job);
    }
    return hipGetLastError();
}

} // namespace xg
