// cuTree: how much of each 8x8 block's cost is inherited by the blocks it predicts from.
//
// Replaces the two loops of Lookahead::estimateCUPropagate
// (source/encoder/slicetype.cpp:3787-3870): primitives.propagateCost
// (estimateCUPropagateCost, source/common/pixel.cpp:906-932) and the scatter of each
// block's amount along its motion vectors into the reference frames' propagateCost.
//
// The reference adds into uint16 cells with saturation at 65535.  Every addend is
// non-negative, so the saturated running sum equals min(total, 65535) whatever the
// order: cells are kept as 64-bit totals, added to with atomics, and read back
// through the clamp.  The double-precision arithmetic of the amount is done with
// separately rounded multiply / add / divide, as the scalar C code does.
#include "device_common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace xg {

__device__ __forceinline__ void clip_add(unsigned long long* cell, int x)
{
    if (x > 0)
        atomicAdd(cell, (unsigned long long)min(x, 65535));
}

__device__ __forceinline__ void propagate_block(const PropagateJob& job)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= job.W * job.H)
        return;
    const int intra = job.intraCost[i];
    const int packed = job.lowresCosts[i];
    const int inter = min(intra, packed & 0x3FFF);
    const unsigned long long inherited = job.propagateIn ? min(job.propagateIn[i], 65535ull) : 0ull;
    const double ownCost = (double)(intra * job.invQscale[i]);
    const double amountF = (double)inherited + ownCost * job.fps;
    const double scaled = amountF * (double)(intra - inter);
    const int amount = (int)(scaled / (double)intra + 0.5);
    if (amount <= 0)
        return;                                     // intra blocks pass nothing on
    const int listsUsed = packed >> 14;
    const int bx = i % job.W, by = i / job.W;
#pragma unroll
    for (int list = 0; list < 2; list++)
    {
        if (!((listsUsed >> list) & 1))
            continue;
        int share = amount;
        if (listsUsed == 3)
            share = (share * (list ? 64 - job.bipredWeight : job.bipredWeight) + 32) >> 6;
        unsigned long long* ref = job.refCosts[list];
        int x = job.mvs[list][2 * i], y = job.mvs[list][2 * i + 1];
        if (!(x | y))
        {
            clip_add(ref + i, share);
            continue;
        }
        // the vector is in quarter pixels of the half-res plane: 32 units per block; spread over
        // the four blocks the displaced block overlaps, by area
        const int cx = (x >> 5) + bx, cy = (y >> 5) + by;
        x &= 31;
        y &= 31;
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int tx = cx + (k & 1), ty = cy + (k >> 1);
            const int w = ((k >> 1) ? y : 32 - y) * ((k & 1) ? x : 32 - x);
            if (tx >= 0 && tx < job.W && ty >= 0 && ty < job.H)
                clip_add(ref + ty * job.W + tx, (share * w + 512) >> 10);
        }
    }
}

__global__ void __launch_bounds__(256) cutree_propagate_kernel(PropagateJob job)
{
    propagate_block(job);
}

// several frames whose passes do not depend on each other, in one launch: blockIdx.y = frame
__global__ void __launch_bounds__(256) cutree_propagate_batch_kernel(const PropagateJob* jobs)
{
    const PropagateJob job = jobs[blockIdx.y];
    propagate_block(job);
}

hipError_t launch_cutree_propagate_batch(hipStream_t s, const PropagateJob* dJobs, int count, int W, int H)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(cutree_propagate_batch_kernel, dim3((W * H + 255) / 256, count), dim3(256), 0, s, dJobs);
    return hipGetLastError();
}

// cells[k][0 .. counts[k]) = 0 for each array of the list
__global__ void __launch_bounds__(256) cutree_clear_batch_kernel(unsigned long long* const* cells, const int* counts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < counts[blockIdx.y])
        cells[blockIdx.y][i] = 0;
}

hipError_t launch_cutree_clear_batch(hipStream_t s, unsigned long long* const* dCells, const int* dCounts, int arrays, int maxCount)
{
    if (arrays <= 0 || maxCount <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(cutree_clear_batch_kernel, dim3((maxCount + 255) / 256, arrays), dim3(256), 0, s, dCells, dCounts);
    return hipGetLastError();
}

hipError_t launch_cutree_propagate(hipStream_t s, const PropagateJob& job)
{
    const int n = job.W * job.H;
    hipLaunchKernelGGL(cutree_propagate_kernel, dim3((n + 255) / 256), dim3(256), 0, s, job);
    return hipGetLastError();
}

} // namespace xg
