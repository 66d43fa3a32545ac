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
// Sixteen lanes per block: lane k owns luma row k (16 pixels) and one chroma row
// (lanes 0..7 the U rows, 8..15 the V rows); a wave measures four neighbouring
// blocks, which read 64 contiguous luma pixels per row.  Row sums are reduced with
// DPP inside the 16-lane row.  Samples beyond the picture are the replicated edge,
// as PicYuv::copyFromPicture pads them (source/common/picyuv.cpp:340-400).
#include "device_common.h"
#include "kernels.h"

namespace xg {

// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ uint32_t row_sum16(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);   // row_mirror
    return v;
}

template <typename P>
__global__ void __launch_bounds__(256) aq_energy_kernel(AqJob job)
{
    const int lane = threadIdx.x & 63, k = lane & 15;
    const int blocksX = (job.width + 15) >> 4, blocksY = (job.height + 15) >> 4;
    const int bx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);     // four waves x four blocks
    const int by = blockIdx.y;
    const bool live = bx < blocksX;
    const int cbx = min(bx, blocksX - 1);

    // luma row k of the block
    uint32_t sumY = 0, sqrY = 0;
    {
        const int y = min(16 * by + k, job.height - 1);
        const P* row = (const P*)job.luma + (long long)y * job.lumaStride;
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            const uint32_t v = row[min(16 * cbx + i, job.width - 1)];
            sumY += v;
            sqrY += v * v;
        }
    }
    // chroma: lanes 0..7 row k of U, lanes 8..15 row k - 8 of V
    uint32_t sumC = 0, sqrC = 0;
    {
        const int cw = job.width >> 1, ch = job.height >> 1;
        const P* plane = (const P*)(k < 8 ? job.u : job.v);
        const int y = min(8 * by + (k & 7), ch - 1);
        const P* row = plane + (long long)y * job.chromaStride;
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const uint32_t v = row[min(8 * cbx + i, cw - 1)];
            sumC += v;
            sqrC += v * v;
        }
    }
    sumY = row_sum16(sumY);
    sqrY = row_sum16(sqrY);
    // U total in lanes 0..7, V total in lanes 8..15: reduce inside each half row only
    uint32_t sC = sumC, qC = sqrC;
    sC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sC, 0xB1, 0xf, 0xf, true);
    sC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sC, 0x4E, 0xf, 0xf, true);
    sC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sC, 0x141, 0xf, 0xf, true);
    qC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qC, 0xB1, 0xf, 0xf, true);
    qC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qC, 0x4E, 0xf, 0xf, true);
    qC += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qC, 0x141, 0xf, 0xf, true);
    // acEnergyVar: ssd - (sum * sum >> shift), shift 8 for 16x16 and 6 for 8x8
    const uint32_t eY = sqrY - (uint32_t)(((unsigned long long)sumY * sumY) >> 8);
    const uint32_t eC = qC - (uint32_t)(((unsigned long long)sC * sC) >> 6);
    // the V energy sits in lanes 8..15: bring it to lane 0's side with the row mirror (lane i <-> 15 - i)
    const uint32_t eV = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)eC, 0x140, 0xf, 0xf, true);
    if (live && k == 0)
    {
        job.energy[(long long)by * blocksX + bx] = eY + eC + eV;
        if (job.lumaSum)
            job.lumaSum[(long long)by * blocksX + bx] = sumY;
    }
    // frame sums: lane 0 of a block carries Y and U, lane 8 carries V
    if (live && (k == 0 || k == 8))
    {
        if (k == 0)
        {
            atomicAdd(job.wp + 0, (unsigned long long)sumY);
            atomicAdd(job.wp + 3, (unsigned long long)sqrY);
            atomicAdd(job.wp + 1, (unsigned long long)sC);
            atomicAdd(job.wp + 4, (unsigned long long)qC);
        }
        else
        {
            atomicAdd(job.wp + 2, (unsigned long long)sC);
            atomicAdd(job.wp + 5, (unsigned long long)qC);
        }
    }
    (void)blocksY;
}

hipError_t launch_aq_energy(hipStream_t s, int bpp, const AqJob& job)
{
    const int blocksX = (job.width + 15) >> 4, blocksY = (job.height + 15) >> 4;
    dim3 grid((blocksX + 15) / 16, blocksY);
    if (bpp == 1)
        hipLaunchKernelGGL(aq_energy_kernel<uint8_t>, grid, dim3(256), 0, s, job);
    else
        hipLaunchKernelGGL(aq_energy_kernel<uint16_t>, grid, dim3(256), 0, s, job);
    return hipGetLastError();
}

} // namespace xg
