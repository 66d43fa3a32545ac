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
template <typename P, int B>
__global__ void __launch_bounds__(256) aq_energy_kernel(AqJob job)
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

} // namespace xg
