// Two formulations of the 8x8 SATD pass of the lowres block search on gfx950, side by side: what a wave pays to
// compare its encode block with eight candidate blocks (satd8<8,8>, source/common/pixel.cpp:210-289).
//
//   valu: the search kernel's own form (device_common.h): a lane holds a row of its group's candidate, packed 16-bit
//         butterflies, the vertical transform across a quad of lanes by DPP, the last butterfly as a maximum.
//   mfma: the Hadamard as a matrix product on the matrix cores.  The transform is linear, so H(fenc - ref) =
//         H(fenc - 128) - H(ref - 128) and both terms are products of the 16x16 matrix H4 (x) H4 with pixels biased
//         into int8: one v_mfma_i32_16x16x32_i8 transforms sixteen 4x4 blocks, its 32-deep inner dimension holding
//         the sixteen reference pixels of a block (matrix entries -h) and the sixteen encode pixels (+h).  Two
//         instructions cover the 32 4x4 blocks of eight candidates; what remains on the vector unit is the biasing,
//         two lane-half swaps that put the encode pixels beside the reference pixels, the absolute sums of the
//         coefficients and the sums across lane rows.
//
// Both read their operands in the lane layout that suits them (the search kernel stages candidates in LDS and can
// read either), run REPS passes whose inputs change from pass to pass, and are checked against a plain C SATD.
// Timed two ways: `throughput` (passes independent, 1 and 8 waves per SIMD) and `latency` (each pass's input
// depends on the previous result, one wave per SIMD).  Prints one JSON object.
#include "../../x265_amd/csrc/device_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace xg;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int REPS = 2048;
typedef int v4i __attribute__((ext_vector_type(4)));

// ---- plain C reference: sum of |4x4 Hadamard| over the two 4x4 blocks of each 8x4 half, halved per half ----
static int satd8x8_host(const uint8_t* a, const uint8_t* b)
{
    int total = 0;
    for (int half = 0; half < 2; half++)
    {
        int sum = 0;
        for (int blk = 0; blk < 2; blk++)
        {
            int d[4][4], t[4][4];
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++)
                    d[y][x] = (int)a[(4 * half + y) * 8 + 4 * blk + x] - (int)b[(4 * half + y) * 8 + 4 * blk + x];
            for (int y = 0; y < 4; y++)
            {
                const int s01 = d[y][0] + d[y][1], d01 = d[y][0] - d[y][1], s23 = d[y][2] + d[y][3], d23 = d[y][2] - d[y][3];
                t[y][0] = s01 + s23; t[y][1] = d01 + d23; t[y][2] = s01 - s23; t[y][3] = d01 - d23;
            }
            for (int x = 0; x < 4; x++)
            {
                const int s01 = t[0][x] + t[1][x], d01 = t[0][x] - t[1][x], s23 = t[2][x] + t[3][x], d23 = t[2][x] - t[3][x];
                sum += abs(s01 + s23) + abs(d01 + d23) + abs(s01 - s23) + abs(d01 - d23);
            }
        }
        total += sum >> 1;
    }
    return total;
}

// Pass `rep` perturbs the candidates (a byte-wise xor keeps pixels in range) so that no pass can be hoisted; with
// CHAIN the perturbation also depends on the previous pass's result.
template <bool CHAIN> __device__ __forceinline__ uint32_t perturb(int rep, int prev)
{
    const uint32_t v = CHAIN ? (uint32_t)(rep + (prev & 1)) : (uint32_t)rep;
    return (v & 0x3fu) * 0x01010101u;
}

// in: per wave 64 encode bytes then 8 x 64 candidate bytes, row-major 8x8
template <bool CHAIN>
__global__ void __launch_bounds__(256) satd_valu_kernel(const uint8_t* in, int* out, unsigned long long* ticks)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, g = lane >> 3, r = lane & 7;
    const uint8_t* base = in + (size_t)wave * 576;
    const Row<uint8_t> fr = load_row<uint8_t>(base + 8 * r);
    const Cols f = pair_cols<uint8_t>(fr);
    const Row<uint8_t> ref0 = load_row<uint8_t>(base + 64 + 64 * g + 8 * r);
    int acc = 0, prev = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < REPS; rep++)
    {
        const uint32_t x = perturb<CHAIN>(rep, prev);
        Row<uint8_t> ref;
        ref.w[0] = ref0.w[0] ^ x;
        ref.w[1] = ref0.w[1] ^ x;
        prev = satd8x8<uint8_t>(f, ref, r);
        acc += prev;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r == 0)
        out[wave * 8 + g] = acc;
    if (lane == 0)
        ticks[wave] = t1 - t0;
}

// The same pass on the matrix cores.  Lane l = 16 kb + j: j = 4 c' + q is 4x4 block q (bit 1: lower half, bit 0:
// right half) of candidate c' (lanes 0..31: candidates 0..3, lanes 32..63: candidates 4..7); the lane holds rows
// 2 (kb & 1) and 2 (kb & 1) + 1 of that block, four pixels each, as two dwords.
template <bool CHAIN>
__global__ void __launch_bounds__(256) satd_mfma_kernel(const uint8_t* in, int* out, unsigned long long* ticks)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint8_t* base = in + (size_t)wave * 576;
    const int kb = lane >> 4, j = lane & 15, q = j & 3;
    const int cand = (j >> 2) + ((lane >> 5) << 2);
    const int row0 = 4 * (q >> 1) + 2 * (kb & 1), col0 = 4 * (q & 1);
    uint32_t r0, r1, f0, f1;
    __builtin_memcpy(&r0, base + 64 + 64 * cand + 8 * row0 + col0, 4);
    __builtin_memcpy(&r1, base + 64 + 64 * cand + 8 * (row0 + 1) + col0, 4);
    __builtin_memcpy(&f0, base + 8 * row0 + col0, 4);
    __builtin_memcpy(&f1, base + 8 * (row0 + 1) + col0, 4);
    f0 ^= 0x80808080u;      // the encode block is biased once per block search
    f1 ^= 0x80808080u;
    // matrix operand: row i = lane & 15 of H4 (x) H4 (natural order), entries 8 kb .. 8 kb + 7 of its 32: pixels
    // (2 (kb & 1) + t / 4, t % 4) of a block, negated for the reference half (kb < 2)
    uint32_t a0 = 0, a1 = 0;
    {
        const int i = lane & 15, u = i >> 2, v = i & 3;
        for (int t = 0; t < 8; t++)
        {
            const int py = 2 * (kb & 1) + (t >> 2), px = t & 3;
            int s = ((__builtin_popcount(u & py) + __builtin_popcount(v & px)) & 1) ? -1 : 1;
            if (kb < 2)
                s = -s;
            const uint32_t byte = (uint32_t)(s & 0xff);
            if (t < 4)
                a0 |= byte << (8 * t);
            else
                a1 |= byte << (8 * (t - 4));
        }
    }
    const long A = (long)(((unsigned long long)a1 << 32) | a0);
    int acc = 0, prev = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < REPS; rep++)
    {
        const uint32_t x = perturb<CHAIN>(rep, prev) ^ 0x80808080u;
        // lanes 32..63 trade their candidate pixels for the encode pixels: B1 = [cands 0..3 | fenc], B2 = [cands 4..7 | fenc]
        const auto s0 = __builtin_amdgcn_permlane32_swap(r0 ^ x, f0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(r1 ^ x, f1, false, false);
        const long B1 = (long)(((unsigned long long)s1[0] << 32) | s0[0]);
        const long B2 = (long)(((unsigned long long)s1[1] << 32) | s0[1]);
        const v4i zero = { 0, 0, 0, 0 };
        const v4i d1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, B1, zero, 0, 0, 0);
        const v4i d2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, B2, zero, 0, 0, 0);
        // four coefficients per lane and product: packed to 16 bits (|c| <= 16 * 255), absolute values, summed
        uint32_t sum1, sum2;
        {
            const s16x2 p0 = as_s16x2(__builtin_amdgcn_perm((uint32_t)d1[1], (uint32_t)d1[0], 0x05040100u));
            const s16x2 p1 = as_s16x2(__builtin_amdgcn_perm((uint32_t)d1[3], (uint32_t)d1[2], 0x05040100u));
            const s16x2 z = { 0, 0 };
            sum1 = __builtin_amdgcn_sad_u16(as_u32(__builtin_elementwise_max(p0, z - p0)), 0u, 0u);
            sum1 = __builtin_amdgcn_sad_u16(as_u32(__builtin_elementwise_max(p1, z - p1)), 0u, sum1);
            const s16x2 p2 = as_s16x2(__builtin_amdgcn_perm((uint32_t)d2[1], (uint32_t)d2[0], 0x05040100u));
            const s16x2 p3 = as_s16x2(__builtin_amdgcn_perm((uint32_t)d2[3], (uint32_t)d2[2], 0x05040100u));
            sum2 = __builtin_amdgcn_sad_u16(as_u32(__builtin_elementwise_max(p2, z - p2)), 0u, 0u);
            sum2 = __builtin_amdgcn_sad_u16(as_u32(__builtin_elementwise_max(p3, z - p3)), 0u, sum2);
        }
        // a block's sixteen coefficients sit in four lane rows: add the halves (both products at once), then the rows
        const auto h = __builtin_amdgcn_permlane32_swap(sum1, sum2, false, false);
        const uint32_t u = h[0] + h[1];                  // lanes 0..31: product 1, lanes 32..63: product 2
        const auto w = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        int t = (int)(w[0] + w[1]);                      // every lane: sum of |coefficients| of block j of its candidate
        t += dpp<DPP_XOR1>(t);                           // left + right 4x4: an 8x4 half
        t >>= 1;
        t += dpp<DPP_XOR2>(t);                           // upper + lower half
        prev = t;
        acc += prev;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // lanes 0..15 hold candidates 0..3 (a quad each), lanes 32..47 candidates 4..7
    if ((lane & 3) == 0 && (lane & 16) == 0)
        out[wave * 8 + (j >> 2) + ((lane >> 5) << 2)] = acc;
    if (lane == 0)
        ticks[wave] = t1 - t0;
}

template <typename K> static int run(K kernel, int wavesPerSimd, int cus, const uint8_t* dIn, int* dOut, unsigned long long* dTicks,
                                     std::vector<int>& hOut, double& ticksPerWave)
{
    const int waves = cus * 4 * wavesPerSimd, blocks = waves / 4;
    for (int warm = 0; warm < 3; warm++)
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, dIn, dOut, dTicks);
    CHK(hipDeviceSynchronize());
    hOut.resize((size_t)waves * 8);
    CHK(hipMemcpy(hOut.data(), dOut, sizeof(int) * hOut.size(), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> t(waves);
    CHK(hipMemcpy(t.data(), dTicks, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
    std::nth_element(t.begin(), t.begin() + waves / 2, t.end());
    ticksPerWave = (double)t[waves / 2];
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, maxWaves = cus * 32;
    std::vector<uint8_t> hIn((size_t)maxWaves * 576);
    unsigned s = 12345;
    for (auto& b : hIn)
    {
        s = s * 1664525u + 1013904223u;
        b = (uint8_t)(s >> 24);
    }
    // a few extreme blocks: all-min against all-max, equal blocks
    for (int i = 0; i < 64; i++) { hIn[i] = 0; hIn[64 + i] = 255; hIn[128 + i] = 0; }
    uint8_t* dIn;
    int* dOut;
    unsigned long long* dTicks;
    CHK(hipMalloc((void**)&dIn, hIn.size()));
    CHK(hipMalloc((void**)&dOut, sizeof(int) * (size_t)maxWaves * 8));
    CHK(hipMalloc((void**)&dTicks, sizeof(unsigned long long) * maxWaves));
    CHK(hipMemcpy(dIn, hIn.data(), hIn.size(), hipMemcpyHostToDevice));
    // expected sums over the REPS passes (independent form) for the first waves
    const int checkWaves = 64;
    std::vector<long long> want((size_t)checkWaves * 8, 0);
    for (int w = 0; w < checkWaves; w++)
        for (int g = 0; g < 8; g++)
            for (int rep = 0; rep < REPS; rep++)
            {
                uint8_t ref[64];
                const uint8_t x = (uint8_t)(rep & 0x3f);
                for (int i = 0; i < 64; i++)
                    ref[i] = hIn[(size_t)w * 576 + 64 + 64 * g + i] ^ x;
                want[(size_t)w * 8 + g] += satd8x8_host(&hIn[(size_t)w * 576], ref);
            }
    std::vector<int> hv, hm;
    double tv1, tv8, tm1, tm8, tvc, tmc;
    if (run(satd_valu_kernel<false>, 1, cus, dIn, dOut, dTicks, hv, tv1)) return 1;
    if (run(satd_mfma_kernel<false>, 1, cus, dIn, dOut, dTicks, hm, tm1)) return 1;
    int badV = 0, badM = 0;
    for (int i = 0; i < checkWaves * 8; i++)
    {
        badV += hv[i] != (int)want[i];
        badM += hm[i] != (int)want[i];
    }
    if (run(satd_valu_kernel<false>, 8, cus, dIn, dOut, dTicks, hv, tv8)) return 1;
    if (run(satd_mfma_kernel<false>, 8, cus, dIn, dOut, dTicks, hm, tm8)) return 1;
    int badCross = 0;
    for (size_t i = 0; i < hv.size(); i++)
        badCross += hv[i] != hm[i];
    if (run(satd_valu_kernel<true>, 1, cus, dIn, dOut, dTicks, hv, tvc)) return 1;
    if (run(satd_mfma_kernel<true>, 1, cus, dIn, dOut, dTicks, hm, tmc)) return 1;
    int badChain = 0;
    for (size_t i = 0; i < hv.size(); i++)
        badChain += hv[i] != hm[i];
    printf("{\"device\": \"%s\", \"passes_per_wave\": %d, \"mismatches\": {\"valu_vs_c\": %d, \"mfma_vs_c\": %d, "
           "\"mfma_vs_valu_all_waves\": %d, \"mfma_vs_valu_chained\": %d}, "
           "\"cycles_per_pass_per_simd\": {\"valu\": {\"1_wave\": %.1f, \"8_waves\": %.1f, \"1_wave_chained\": %.1f}, "
           "\"mfma\": {\"1_wave\": %.1f, \"8_waves\": %.1f, \"1_wave_chained\": %.1f}}}\n",
           prop.gcnArchName, REPS, badV, badM, badCross, badChain, tv1 / REPS, tv8 / REPS / 8, tvc / REPS, tm1 / REPS,
           tm8 / REPS / 8, tmc / REPS);
    return (badV || badM || badCross || badChain) ? 2 : 0;
}
