// Block energies for adaptive quantisation and the frame sums weighted prediction starts from.
//
// Replaces the pixel passes of LookaheadTLD::calcAdaptiveQuantFrame
// (source/encoder/slicetype.cpp:454-704): acEnergyCu over every 16x16 block of the
// source picture (:266-285) = pixel_var of the 16x16 luma block and of the two 8x8
// chroma blocks of a 4:2:0 picture (source/common/pixel.cpp:695-712), each turned
// into an AC energy by acEnergyVar (slicetype.cpp:51-59), which also accumulates the
// per-plane sums Lowres::wp_sum / wp_ssd.  The floating-point mapping from energy to
// QP offsets (log2 / pow, :560-680) stays on the host.
//
// With 16x16 blocks, sixteen lanes per block: lane k owns luma row k (16 pixels) and
// one chroma row (lanes 0..7 the U rows, 8..15 the V rows); a wave measures four
// neighbouring blocks, which read 64 contiguous luma pixels per row.  With qgSize 8 the
// blocks are 8x8 luma + 4x4 chroma, eight lanes each, eight blocks per wave.  Row sums
// are reduced with DPP inside the block's lanes.  Samples beyond the picture are the replicated edge,
// as PicYuv::copyFromPicture pads them (source/common/picyuv.cpp:340-400).
#include "device_common.h"
#include "kernels.h"

namespace xg {

// B = 16 (qgSize >= 16: 16x16 luma, 8x8 chroma, shifts 8 / 6) or 8 (qgSize 8: 8x8 luma, 4x4 chroma, shifts 6 / 4;
// slicetype.cpp:62-86).  B lanes per block, 64 / B blocks per wave.  The block loop runs while
// blockX < visitWidth and blockY < visitHeight (normally the picture size; the variance tail of
// calcAdaptiveQuantFrame walks a picture rounded to 16, :674-675, :692-694); results are stored in loop order.
// sum and sum of squares of N consecutive pixels (N * sizeof(P) a multiple of 8 bytes), read as 8-byte pieces
template <typename P, int N>
__device__ __forceinline__ void sum_pixels(const P* p, uint32_t& sum, uint32_t& sqr)
{
    constexpr int PIECES = N * (int)sizeof(P) / 8;
#pragma unroll
    for (int q = 0; q < PIECES; q++)
    {
        const u32x2_unaligned v = *(const u32x2_unaligned*)((const char*)p + 8 * q);
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const uint32_t w = h ? v.y : v.x;
            if (sizeof(P) == 1)
            {
                const uint32_t b0 = w & 0xffu, b1 = (w >> 8) & 0xffu, b2 = (w >> 16) & 0xffu, b3 = w >> 24;
                sum = __builtin_amdgcn_sad_u8(w, 0u, sum);
                sqr += b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3;
            }
            else
            {
                const uint32_t lo = w & 0xffffu, hi = w >> 16;
                sum += lo + hi;
                sqr += lo * lo + hi * hi;
            }
        }
    }
}

template <typename P, int B>
__device__ __forceinline__ void aq_energy_block_rows(const AqJob& job)
{
    constexpr int PER_WAVE = 64 / B, HALF = B / 2;
    const int lane = threadIdx.x & 63, k = lane & (B - 1);
    const int blocksX = (job.visitWidth + B - 1) / B;
    const int bx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PER_WAVE + lane / B;     // four waves x 64 / B blocks
    const int by = blockIdx.y;
    const bool live = bx < blocksX;
    const int cbx = min(bx, blocksX - 1);

    // luma row k of the block
    uint32_t sumY = 0, sqrY = 0;
    {
        const int y = min(B * by + k, job.height - 1);
        const P* row = (const P*)job.luma + (long long)y * job.lumaStride;
        if (B * cbx + B <= job.width)                 // the whole row segment lies inside the picture: wide loads
            sum_pixels<P, B>(row + B * cbx, sumY, sqrY);
        else
#pragma unroll
            for (int i = 0; i < B; i++)
            {
                const uint32_t v = row[min(B * cbx + i, job.width - 1)];
                sumY += v;
                sqrY += v * v;
            }
    }
    // chroma: the lower half of the lanes take the U rows, the upper half the V rows
    uint32_t sC = 0, qC = 0;
    {
        const int cw = job.width >> 1, ch = job.height >> 1;
        const P* plane = (const P*)(k < HALF ? job.u : job.v);
        const int y = min(HALF * by + (k & (HALF - 1)), ch - 1);
        const P* row = plane + (long long)y * job.chromaStride;
        if (HALF * sizeof(P) >= 8 && HALF * cbx + HALF <= cw)
            sum_pixels<P, (HALF * sizeof(P) >= 8 ? HALF : 8)>(row + HALF * cbx, sC, qC);
        else
#pragma unroll
            for (int i = 0; i < HALF; i++)
            {
                const uint32_t v = row[min(HALF * cbx + i, cw - 1)];
                sC += v;
                qC += v * v;
            }
    }
    // luma over the B lanes of the block; U total in the lower half, V total in the upper half
    sumY += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sumY, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    sumY += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sumY, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    sumY += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sumY, 0x141, 0xf, 0xf, true);   // row_half_mirror
    sqrY += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sqrY, 0xB1, 0xf, 0xf, true);
    sqrY += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sqrY, 0x4E, 0xf, 0xf, true);
    sqrY += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sqrY, 0x141, 0xf, 0xf, true);
    sC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sC, 0xB1, 0xf, 0xf, true);
    sC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sC, 0x4E, 0xf, 0xf, true);
    qC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qC, 0xB1, 0xf, 0xf, true);
    qC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qC, 0x4E, 0xf, 0xf, true);
    if (B == 16)
    {
        sumY += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sumY, 0x140, 0xf, 0xf, true);   // row_mirror
        sqrY += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sqrY, 0x140, 0xf, 0xf, true);
        sC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sC, 0x141, 0xf, 0xf, true);
        qC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qC, 0x141, 0xf, 0xf, true);
    }
    // acEnergyVar: ssd - (sum * sum >> shift)
    const uint32_t eY = sqrY - (uint32_t)(((unsigned long long)sumY * sumY) >> (B == 16 ? 8 : 6));
    const uint32_t eC = qC - (uint32_t)(((unsigned long long)sC * sC) >> (B == 16 ? 6 : 4));
    // the V energy sits in the upper half: the mirror of the block's lanes (lane i <-> B - 1 - i) hands it to lane 0
    const uint32_t eV = B == 16 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)eC, 0x140, 0xf, 0xf, true)
                                : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)eC, 0x141, 0xf, 0xf, true);
    if (live && k == 0)
    {
        job.energy[(long long)by * blocksX + bx] = eY + eC + eV;
        if (job.lumaSum)
            job.lumaSum[(long long)by * blocksX + bx] = sumY;
    }
    // Frame sums: lane 0 of a block carries Y and U, the first lane of the upper half carries V.  They are added up
    // over the wave and the workgroup first, and the workgroup adds its six totals to one of AQ_WP_SHARDS sets of
    // accumulators, each on a 128-byte line of its own: thousands of atomics on one line serialise (they took
    // four fifths of this kernel's time), a few per line do not.  The host adds the shards (integer sums, so the
    // order does not matter).
    const bool first = live && k == 0, upper = live && k == HALF;
    unsigned long long part[6] = { first ? sumY : 0u, first ? sC : 0u, upper ? sC : 0u,
                                   first ? sqrY : 0u, first ? qC : 0u, upper ? qC : 0u };
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
            part[i] += __shfl_xor(part[i], m, 64);
    __shared__ unsigned long long red[4][6];
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 6; i++)
            red[threadIdx.x >> 6][i] = part[i];
    __syncthreads();
    if (threadIdx.x < 6)
    {
        const unsigned long long total = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const unsigned shard = (blockIdx.x + blockIdx.y * gridDim.x) % AQ_WP_SHARDS;
        atomicAdd(job.wp + shard * AQ_WP_SHARD_STRIDE + threadIdx.x, total);
    }
}

template <typename P, int B>
__global__ void __launch_bounds__(256) aq_energy_kernel(AqJob job)
{
    aq_energy_block_rows<P, B>(job);
}

// the pictures of a batch in one launch: blockIdx.z = picture
template <typename P, int B>
__global__ void __launch_bounds__(256) aq_energy_batch_kernel(const AqJob* jobs)
{
    const AqJob job = jobs[blockIdx.z];
    aq_energy_block_rows<P, B>(job);
}

// dst[i][k] = src[i][k], k < count: the scale factors of a batch of pictures from one staging buffer to their slots
__global__ void __launch_bounds__(256) scatter_i32_kernel(const int32_t* const* src, int32_t* const* dst, int count)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < count)
        dst[blockIdx.y][k] = src[blockIdx.y][k];
}

hipError_t launch_scatter_i32(hipStream_t s, const int32_t* const* dSrc, int32_t* const* dDst, int count, int n)
{
    if (n <= 0 || count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(scatter_i32_kernel, dim3((count + 255) / 256, n), dim3(256), 0, s, dSrc, dDst, count);
    return hipGetLastError();
}

// Quadrant statistics of LookaheadTLD::xPreanalyze (source/encoder/slicetype.cpp:359-398): the picture is cut
// into parts of partW x partH luma samples (clipped at the right / bottom edge), each part into four quadrants
// at half its clipped width and height, and every quadrant gets its sum and sum of squares.  One wave per part;
// out[part * 8 + q] = sum, out[part * 8 + 4 + q] = sum of squares, q = 2 * (lower half) + (right half).
template <typename P>
__global__ void __launch_bounds__(256) aq_quadrant_kernel(const P* luma, long long stride, int width, int height, int partW,
                                                          int partH, int partsX, int parts, unsigned long long* out)
{
    const int lane = threadIdx.x & 63;
    const int part = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (part >= parts)
        return;
    const int x0 = (part % partsX) * partW, y0 = (part / partsX) * partH;
    const int cw = min(partW, width - x0), ch = min(partH, height - y0);
    const int halfW = cw >> 1, halfH = ch >> 1;
    uint32_t sum[4] = { 0, 0, 0, 0 };
    unsigned long long sq[4] = { 0, 0, 0, 0 };
    for (int i = lane; i < cw * ch; i += 64)
    {
        const int r = i / cw, c = i - r * cw;
        const uint32_t v = luma[(long long)(y0 + r) * stride + x0 + c];
        const int q = (r >= halfH ? 2 : 0) + (c >= halfW ? 1 : 0);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            sum[k] += q == k ? v : 0u;
            sq[k] += q == k ? (unsigned long long)(v * v) : 0ull;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        unsigned long long s = sum[k], t = sq[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
        {
            s += __shfl_xor(s, m, 64);
            t += __shfl_xor(t, m, 64);
        }
        if (lane == 0)
        {
            out[(long long)part * 8 + k] = s;
            out[(long long)part * 8 + 4 + k] = t;
        }
    }
}

hipError_t launch_aq_quadrants(hipStream_t s, int bpp, const void* luma, long long stride, int width, int height, int partW,
                               int partH, unsigned long long* out)
{
    const int partsX = (width + partW - 1) / partW, partsY = (height + partH - 1) / partH, parts = partsX * partsY;
    if (parts < 1)
        return hipSuccess;
    dim3 grid((parts + 3) / 4);
    if (bpp == 1)
        hipLaunchKernelGGL(aq_quadrant_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)luma, stride, width, height,
                           partW, partH, partsX, parts, out);
    else
        hipLaunchKernelGGL(aq_quadrant_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)luma, stride, width, height,
                           partW, partH, partsX, parts, out);
    return hipGetLastError();
}

// n pictures of one size with 16x16 blocks, jobs on the device
hipError_t launch_aq_energy_batch(hipStream_t s, int bpp, const AqJob* dJobs, int n, int visitWidth, int visitHeight)
{
    const int blocksX = (visitWidth + 15) / 16, blocksY = (visitHeight + 15) / 16;
    if (blocksX < 1 || blocksY < 1 || n < 1)
        return hipSuccess;
    dim3 grid((blocksX + 15) / 16, blocksY, n);
    if (bpp == 1)
        hipLaunchKernelGGL((aq_energy_batch_kernel<uint8_t, 16>), grid, dim3(256), 0, s, dJobs);
    else
        hipLaunchKernelGGL((aq_energy_batch_kernel<uint16_t, 16>), grid, dim3(256), 0, s, dJobs);
    return hipGetLastError();
}

hipError_t launch_aq_energy(hipStream_t s, int bpp, const AqJob& job)
{
    const int B = job.blockSize == 8 ? 8 : 16;
    const int blocksX = (job.visitWidth + B - 1) / B, blocksY = (job.visitHeight + B - 1) / B;
    const int perGroup = 256 / B;
    dim3 grid((blocksX + perGroup - 1) / perGroup, blocksY);
    if (blocksX < 1 || blocksY < 1)
        return hipSuccess;
    if (bpp == 1)
    {
        if (B == 16)
            hipLaunchKernelGGL((aq_energy_kernel<uint8_t, 16>), grid, dim3(256), 0, s, job);
        else
            hipLaunchKernelGGL((aq_energy_kernel<uint8_t, 8>), grid, dim3(256), 0, s, job);
    }
    else
    {
        if (B == 16)
            hipLaunchKernelGGL((aq_energy_kernel<uint16_t, 16>), grid, dim3(256), 0, s, job);
        else
            hipLaunchKernelGGL((aq_energy_kernel<uint16_t, 8>), grid, dim3(256), 0, s, job);
    }
    return hipGetLastError();
}

// ---- edge mode -------------------------------------------------------------------------
// edgeFilter (source/encoder/slicetype.cpp:191-225) and edgeDensityCu's two block sums (:227-263) in one pass:
// a workgroup takes a 16x16 tile of the picture, stages the 22x22 source pixels around it, blurs the 18x18 it needs
// (gaussianBlur5x5, :164-189: the two-pixel frame of the picture keeps source pixels) and applies computeEdge's
// Sobel pair (:100-162) to its own pixels.  The edge picture is the source on the one-pixel frame, the threshold
// test inside, zero beyond the picture; the angle picture is zero on the frame and beyond.  The threshold test on
// sqrtf(gH^2 + gV^2) is done on the integer sum of squares: both thresholds' squares are below 2^24, where the
// float sum is exact and sqrtf is monotonic.
template <typename P>
__global__ void __launch_bounds__(256) aq_edge_kernel(AqEdgeJob job, int threshold)
{
    __shared__ int src[22][23];
    __shared__ int blur[18][19];
    __shared__ uint32_t acc[4][3];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * 16, y0 = (job.tileRow0 + blockIdx.y) * 16;
    const int w = job.width, h = job.height;
    if (tid < 12)
        acc[tid / 3][tid % 3] = 0;
    for (int i = tid; i < 22 * 22; i += 256)
    {
        const int r = i / 22, c = i - r * 22;
        const int y = min(max(y0 + r - 3, 0), h - 1), x = min(max(x0 + c - 3, 0), w - 1);
        src[r][c] = (int)((const P*)job.luma)[(long long)y * job.lumaStride + x];
    }
    __syncthreads();
    for (int i = tid; i < 18 * 18; i += 256)
    {
        const int r = i / 18, c = i - r * 18;
        const int y = y0 + r - 1, x = x0 + c - 1;           // picture position of blur[r][c]; src[r + 2][c + 2] is its centre
        int v = src[r + 2][c + 2];
        if (y >= 2 && y < h - 2 && x >= 2 && x < w - 2)
        {
            const int s0 = 2 * src[r][c] + 4 * src[r][c + 1] + 5 * src[r][c + 2] + 4 * src[r][c + 3] + 2 * src[r][c + 4];
            const int s1 = 4 * src[r + 1][c] + 9 * src[r + 1][c + 1] + 12 * src[r + 1][c + 2] + 9 * src[r + 1][c + 3] + 4 * src[r + 1][c + 4];
            const int s2 = 5 * src[r + 2][c] + 12 * src[r + 2][c + 1] + 15 * src[r + 2][c + 2] + 12 * src[r + 2][c + 3] + 5 * src[r + 2][c + 4];
            const int s3 = 4 * src[r + 3][c] + 9 * src[r + 3][c + 1] + 12 * src[r + 3][c + 2] + 9 * src[r + 3][c + 3] + 4 * src[r + 3][c + 4];
            const int s4 = 2 * src[r + 4][c] + 4 * src[r + 4][c + 1] + 5 * src[r + 4][c + 2] + 4 * src[r + 4][c + 3] + 2 * src[r + 4][c + 4];
            v = (s0 + s1 + s2 + s3 + s4) / 159;
        }
        blur[r][c] = v;
    }
    __syncthreads();
    const int x = x0 + tx, y = y0 + ty;
    uint32_t edge = 0, theta = 0;
    bool listed = false;
    int gV = 0, gH = 0;
    if (x < w && y < h)
    {
        if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1)
        {
            const int r = ty + 1, c = tx + 1;
            gH = -3 * blur[r - 1][c - 1] + 3 * blur[r - 1][c + 1] - 10 * blur[r][c - 1] + 10 * blur[r][c + 1] - 3 * blur[r + 1][c - 1] + 3 * blur[r + 1][c + 1];
            gV = -3 * blur[r - 1][c - 1] - 10 * blur[r - 1][c] - 3 * blur[r - 1][c + 1] + 3 * blur[r + 1][c - 1] + 10 * blur[r + 1][c] + 3 * blur[r + 1][c + 1];
            edge = gH * gH + gV * gV >= threshold * threshold ? (uint32_t)threshold : 0u;
            if (gV == 0)
                theta = gH >= 0 ? 0 : job.theta180;        // atan2(+0, x): +0 for x >= 0, pi for x < 0
            else if (gH == 0)
                theta = gV > 0 ? job.theta90pos : job.theta90neg;
            else
            {
                // the device's atan2f and the host's may differ in the last bits: take the whole degree only when
                // both ends of a generous error interval agree on it
                const float radians = atan2f((float)gV, (float)gH);
                const float slack = fabsf(radians) * 1.5e-6f;
                int deg[2];
#pragma unroll
                for (int k = 0; k < 2; k++)
                {
                    const float rk = k ? radians + slack : radians - slack;
                    float t = (float)((double)(rk * 180.0f) / 3.14159265);
                    if (t < 0)
                        t = 180.0f + t;
                    deg[k] = (int)t;
                }
                if (deg[0] == deg[1])
                    theta = deg[0];
                else
                    listed = true;
            }
        }
        else
            edge = src[ty + 3][tx + 3];
    }
    const int sub = job.blockSize == 8 ? (ty >> 3) * 2 + (tx >> 3) : 0;
    if (edge)
    {
        atomicAdd(&acc[sub][0], edge);
        atomicAdd(&acc[sub][1], edge * edge);
    }
    if (theta)
        atomicAdd(&acc[sub][2], theta);
    const int blocksPerTile = job.blockSize == 8 ? 2 : 1;
    const int bx = blockIdx.x * blocksPerTile + (job.blockSize == 8 ? tx >> 3 : 0);
    const int by = (job.tileRow0 + blockIdx.y) * blocksPerTile + (job.blockSize == 8 ? ty >> 3 : 0);
    if (listed)
    {
        const uint32_t at = atomicAdd(job.listCount, 1u);
        // three words per record: a 12-bit picture's gradients reach 16 * 4095, beyond 16 bits
        job.list[3 * at] = (uint32_t)(by * job.blocksX + bx);
        job.list[3 * at + 1] = (uint32_t)gV;
        job.list[3 * at + 2] = (uint32_t)gH;
    }
    __syncthreads();
    if (tid < 4 * 3)
    {
        const int sb = tid / 3, what = tid % 3;
        const int obx = blockIdx.x * blocksPerTile + (sb & 1), oby = (job.tileRow0 + blockIdx.y) * blocksPerTile + (sb >> 1);
        const bool exists = (job.blockSize == 8 || sb == 0) && obx * job.blockSize < w && oby * job.blockSize < h;
        if (exists)
        {
            uint32_t* out = what == 0 ? job.edgeSum : what == 1 ? job.edgeSsd : job.thetaSum;
            out[oby * job.blocksX + obx] = acc[sb][what];
        }
    }
}

hipError_t launch_aq_edge(hipStream_t s, int bpp, int depth, const AqEdgeJob& job, int tileRows)
{
    dim3 grid((job.width + 15) / 16, tileRows);
    const int threshold = depth > 8 ? 1023 : 255;        // EDGE_THRESHOLD, source/encoder/slicetype.h:69-73
    if (bpp == 1)
        hipLaunchKernelGGL(aq_edge_kernel<uint8_t>, grid, dim3(256), 0, s, job, threshold);
    else
        hipLaunchKernelGGL(aq_edge_kernel<uint16_t>, grid, dim3(256), 0, s, job, threshold);
    return hipGetLastError();
}

// ---- --aq-motion: the per-block displacement of Lookahead::calcMotionAdaptiveQuantFrame ----
// (source/encoder/slicetype.cpp:3614-3635): the length of the block's vector in each list the request's decision
// used (lowresCosts >> 14), halved when both were used.  One thread per 8x8 block over the arrays the estimate left
// on the device; the square roots are IEEE double square roots as on the host, and what follows them (the power of
// 0.1, the frame's mean and deviation, the offsets) is libm and ordered summation and stays on the host.
__global__ void __launch_bounds__(256) aq_motion_kernel(const uint16_t* lowresCosts, const int32_t* mvs0, const int32_t* mvs1,
                                                        int n, double* displacement)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const int lists = lowresCosts[i] >> LOWRES_COST_SHIFT;
    double d = 0;
    if (lists & 1)
    {
        const int x = abs(mvs0[2 * i]), y = abs(mvs0[2 * i + 1]);
        d += __builtin_sqrt((double)(x * x) + (double)(y * y));
    }
    if (lists & 2)
    {
        const int x = abs(mvs1[2 * i]), y = abs(mvs1[2 * i + 1]);
        d += __builtin_sqrt((double)(x * x) + (double)(y * y));
    }
    if (lists == 3)
        d = d / 2;
    displacement[i] = d;
}

hipError_t launch_aq_motion(hipStream_t s, const uint16_t* lowresCosts, const int32_t* mvs0, const int32_t* mvs1, int n,
                            double* displacement)
{
    hipLaunchKernelGGL(aq_motion_kernel, dim3((n + 255) / 256), dim3(256), 0, s, lowresCosts, mvs0, mvs1, n, displacement);
    return hipGetLastError();
}

} // namespace xg
