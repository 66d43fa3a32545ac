// Per-block cost decision with stored vectors, weighted-prediction helpers.
//
// cost_rows_kernel replaces the decision half of
// CostEstimateGroup::estimateCUCost (source/encoder/slicetype.cpp:4427-4431,
// 4495-4556): pick the cheaper list from the stored search costs, for B
// requests also measure the two bidirectional candidates (average of the two
// motion-compensated blocks, average of the co-located blocks), for P requests
// compare against the intra cost; apply the AQ scale; emit lowresCosts and the
// row / frame sums.  No block depends on another here, so every 8-lane group
// of a wave takes its own block (lane = row) and a workgroup owns one block
// row, whose total it reduces in LDS -- sums are exact integers, independent
// of scheduling, with no atomics.
#include "device_common.h"
#include "kernels.h"

namespace xg {

template <typename P>
__global__ void __launch_bounds__(256) cost_rows_kernel(const CostJob* jobs, Geometry g)
{
    __shared__ long long red[4][4];

    // grid (requests, block rows), the request fastest: the requests of a batch move down the picture together, so
    // the rows of a frame that several of them read (as encode frame or as reference) are fetched from memory once
    const CostJob job = jobs[blockIdx.x];
    const int cuY = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 3, r = lane & 7;
    const int W = g.W, H = g.H;
    const int by = 8 * cuY;

    PlaneSet<P> ps0, ps1;
    ps0.origin = (const P*)job.ref0Origin;
    ps0.planeSize = g.planeSize;
    ps0.stride = g.stride;
    ps0.minX = -g.marginX;
    ps0.maxX = g.width + g.marginX - 8;
    ps0.minY = -g.marginY;
    ps0.maxY = g.lines + g.marginY - 8;
    ps1 = ps0;
    ps1.origin = (const P*)job.ref1Origin;

    long long accRow = 0, accEst = 0, accAq = 0, accIntra = 0;
    const int chunks = (W + 7) / 8;
    for (int chunk = wave; chunk < chunks; chunk += 4)
    {
        const int cuXraw = chunk * 8 + grp;
        const bool live = cuXraw < W;
        const int cuX = min(cuXraw, W - 1);
        const int cuXY = cuX + cuY * W;
        int bcost = COST_MAX, listused = 0;
        const int fc0 = job.mvCosts0[cuXY];
        if (fc0 < bcost) { bcost = fc0; listused = 1; }
        if (job.bBidir)
        {
            const int fc1 = job.mvCosts1[cuXY];
            if (fc1 < bcost) { bcost = fc1; listused = 2; }

            // Pixel rows are fetched in a transposed lane layout, lane = (row, block): the eight
            // lanes of a row then read neighbouring blocks, which share cache lines whenever
            // their vectors agree (always for the encode block and the co-located pair), instead
            // of eight different lines per block.  The averages are formed there, and only the
            // three rows the SATD needs are brought to the compute layout (block, row).
            const int lr = grp, lj = r;                       // this lane's row / block while loading
            const int lcuX = min(chunk * 8 + lj, W - 1);
            const int lcuXY = lcuX + cuY * W;
            const int lbx = 8 * lcuX;
            const int m0x = job.mvs0[2 * lcuXY], m0y = job.mvs0[2 * lcuXY + 1];
            const int m1x = job.mvs1[2 * lcuXY], m1y = job.mvs1[2 * lcuXY + 1];
            const bool odd = __any((m0x | m0y | m1x | m1y) & 1) != 0;
            const Row<P> fl = load_row<P>((const P*)job.fencPlane0 + (long long)(by + lr) * g.stride + lbx);
            // average of the two motion-compensated predictions
            const Row<P> biMc = avg_row<P>(mc_row<P>(ps0, lbx, by, lr, m0x, m0y, odd), mc_row<P>(ps1, lbx, by, lr, m1x, m1y, odd));
            // average of the co-located blocks
            const Row<P> biCo = avg_row<P>(fetch_row<P>(ps0, lbx, by, lr, 0, 0), fetch_row<P>(ps1, lbx, by, lr, 0, 0));
            const int from = (r * 8 + grp) * 4;               // compute lane (block grp, row r) <- load lane (row r, block grp)
            Row<P> ft, mt, ct;
#pragma unroll
            for (int i = 0; i < PixT<P>::NW; i++)
            {
                ft.w[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)fl.w[i]);
                mt.w[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)biMc.w[i]);
                ct.w[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)biCo.w[i]);
            }
            const Cols f = pair_cols<P>(ft);
            int bicost = satd8x8<P>(f, mt, r);
            if (bicost < bcost) { bcost = bicost; listused = 3; }
            bicost = satd8x8<P>(f, ct, r);
            if (bicost < bcost) { bcost = bicost; listused = 3; }
            bcost += 4;   // lowresPenalty
        }
        else
        {
            bcost += 4;
            const int ic = job.intraCost[cuXY];
            if (ic < bcost) { bcost = ic; listused = 0; }
        }

        const bool scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
        int bcostAq = bcost;
        if (scored && job.invQscale)
            bcostAq = (bcost * job.invQscale[cuXY] + 128) >> 8;
        if (live && r == 0)
        {
            job.lowresCosts[cuXY] = (uint16_t)(min(bcost, LOWRES_COST_MASK) | (listused << LOWRES_COST_SHIFT));
            accRow += bcostAq;
            if (scored)
            {
                accEst += bcost;
                accAq += bcostAq;
                if (!listused && !job.bBidir)
                    accIntra++;
            }
        }
    }

    for (int o = 32; o > 0; o >>= 1)
    {
        accRow += __shfl_down(accRow, o);
        accEst += __shfl_down(accEst, o);
        accAq += __shfl_down(accAq, o);
        accIntra += __shfl_down(accIntra, o);
    }
    if (lane == 0)
    {
        red[wave][0] = accRow;
        red[wave][1] = accEst;
        red[wave][2] = accAq;
        red[wave][3] = accIntra;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        job.rowSatds[cuY] = (int32_t)(red[0][0] + red[1][0] + red[2][0] + red[3][0]);
        job.rowPartials[3 * cuY + 0] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        job.rowPartials[3 * cuY + 1] = red[0][2] + red[1][2] + red[2][2] + red[3][2];
        job.rowPartials[3 * cuY + 2] = red[0][3] + red[1][3] + red[2][3] + red[3][3];
    }
}

// frame totals of one request: costEst, costEstAq, intraMbs (slicetype.cpp:4541-4544)
__global__ void cost_finalize_kernel(const CostJob* jobs, int H)
{
    const CostJob job = jobs[blockIdx.x];
    long long a = 0, b = 0, c = 0;
    for (int y = threadIdx.x; y < H; y += 64)
    {
        a += job.rowPartials[3 * y];
        b += job.rowPartials[3 * y + 1];
        c += job.rowPartials[3 * y + 2];
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        a += __shfl_down(a, o);
        b += __shfl_down(b, o);
        c += __shfl_down(c, o);
    }
    if (threadIdx.x == 0)
    {
        job.sums[0] = a;
        job.sums[1] = b;
        job.sums[2] = c;
        if (job.batchOut)
        {
            job.batchOut[0] = a;
            job.batchOut[1] = b;
            job.batchOut[2] = c;
        }
    }
}

hipError_t launch_cost(hipStream_t s, const Geometry& g, int bpp, const CostJob* dJobs, int n)
{
    if (n <= 0)
        return hipSuccess;
    dim3 grid(n, g.H);
    if (bpp == 1)
        hipLaunchKernelGGL(cost_rows_kernel<uint8_t>, grid, dim3(256), 0, s, dJobs, g);
    else
        hipLaunchKernelGGL(cost_rows_kernel<uint16_t>, grid, dim3(256), 0, s, dJobs, g);
    hipLaunchKernelGGL(cost_finalize_kernel, dim3(n), dim3(64), 0, s, dJobs, g.H);
    return hipGetLastError();
}

// LookaheadTLD::weightCostLuma (slicetype.cpp:836-849): sum over all 8x8 blocks
// of min(SATD(ref, fenc), intraCost).  The uint32 total wraps like the
// reference's accumulator; one atomic per workgroup.
template <typename P>
__device__ __forceinline__ void weight_cost_rows(const WeightCostJob& job, const Geometry& g)
{
    __shared__ unsigned int red[4];
    const int cuY = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 3, r = lane & 7;
    const int W = g.W;
    unsigned int acc = 0;
    const int chunks = (W + 7) / 8;
    for (int chunk = wave; chunk < chunks; chunk += 4)
    {
        const int cuXraw = chunk * 8 + grp;
        const int cuX = min(cuXraw, W - 1);
        const long long off = (long long)(8 * cuY + r) * g.stride + 8 * cuX;
        const Cols f = pair_cols<P>(load_row<P>((const P*)job.fencPlane0 + off));
        Row<P> ref = load_row<P>((const P*)job.refPlane0 + off);
        if (job.weighted)       // uniform over the launch's y slice
        {
            // weight_pp (source/common/pixel.cpp:510-535) on the row just read
            int v[8];
            unpack_row<P>(ref, v);
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const int val = (int)(short)(v[k] << job.correction);
                v[k] = clampi(((job.w0 * val + job.round) >> job.shift) + job.offset, 0, job.pixelMax);
            }
            ref = pack_row<P>(v);
        }
        const int satd = satd8x8<P>(f, ref, r);
        if (cuXraw < W && r == 0)
            acc += (unsigned int)min(satd, job.intraCost[cuX + cuY * W]);
    }
    for (int o = 32; o > 0; o >>= 1)
        acc += __shfl_down(acc, o);
    if (lane == 0)
        red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(job.out, red[0] + red[1] + red[2] + red[3]);
}

template <typename P>
__global__ void __launch_bounds__(256) weight_cost_kernel(WeightCostJob job, Geometry g)
{
    weight_cost_rows<P>(job, g);
}

// every pair of a batch in one launch: blockIdx.y = pair
template <typename P>
__global__ void __launch_bounds__(256) weight_cost_batch_kernel(const WeightCostJob* jobs, Geometry g)
{
    const WeightCostJob job = jobs[blockIdx.y];
    weight_cost_rows<P>(job, g);
}

hipError_t launch_weight_cost_batch(hipStream_t s, const Geometry& g, int bpp, const WeightCostJob* dJobs, int n)
{
    if (n <= 0)
        return hipSuccess;
    dim3 grid(g.H, n);
    if (bpp == 1)
        hipLaunchKernelGGL(weight_cost_batch_kernel<uint8_t>, grid, dim3(256), 0, s, dJobs, g);
    else
        hipLaunchKernelGGL(weight_cost_batch_kernel<uint16_t>, grid, dim3(256), 0, s, dJobs, g);
    return hipGetLastError();
}

// dst[3 * i + k] = src[i][k]: the three frame totals of n (slot, d0, d1) entries in one pass
__global__ void __launch_bounds__(256) gather_totals_kernel(const long long* const* src, long long* dst, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 3 * n)
        dst[i] = src[i / 3][i % 3];
}

hipError_t launch_gather_totals(hipStream_t s, const long long* const* dSrc, long long* dDst, int n)
{
    if (n <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(gather_totals_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, s, dSrc, dDst, n);
    return hipGetLastError();
}

hipError_t launch_weight_cost(hipStream_t s, const Geometry& g, int bpp, const WeightCostJob* job)
{
    dim3 grid(g.H);
    if (bpp == 1)
        hipLaunchKernelGGL(weight_cost_kernel<uint8_t>, grid, dim3(256), 0, s, *job, g);
    else
        hipLaunchKernelGGL(weight_cost_kernel<uint16_t>, grid, dim3(256), 0, s, *job, g);
    return hipGetLastError();
}

// weight_pp over `count` pixels (source/common/pixel.cpp:510-535):
// clip(((w0 * (pix << (14 - depth)) + round) >> shift) + offset)
template <typename P>
__global__ void __launch_bounds__(256) weight_pp_kernel(const P* src, P* dst, long long spans, int correction,
                                                        int pixelMax, int w0, int round, int shift, int offset)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= spans)
        return;
    int v[8];
    unpack_row<P>(load_row<P>(src + 8 * i), v);
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        const int val = (int)(short)(v[k] << correction);
        v[k] = clampi(((w0 * val + round) >> shift) + offset, 0, pixelMax);
    }
    store_row<P>(dst + 8 * i, pack_row<P>(v));
}

hipError_t launch_weight_planes(hipStream_t s, const Geometry& g, int bpp, int depth, int nPlanes, const void* src,
                                void* dst, int w0, int round, int shift, int offset)
{
    const long long spans = nPlanes * g.planeSize / 8;
    dim3 grid((unsigned)((spans + 255) / 256));
    if (bpp == 1)
        hipLaunchKernelGGL(weight_pp_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)src, (uint8_t*)dst, spans,
                           14 - depth, (1 << depth) - 1, w0, round, shift, offset);
    else
        hipLaunchKernelGGL(weight_pp_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst,
                           spans, 14 - depth, (1 << depth) - 1, w0, round, shift, offset);
    return hipGetLastError();
}

} // namespace xg
