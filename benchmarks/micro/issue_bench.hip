// Issue-rate microbenchmark for gfx950: cycles a SIMD (vector) or a compute unit (scalar) spends per
// wave64 instruction of the kinds the lookahead search kernel is made of, and the clock the chip
// holds while it runs them.  Prints one JSON object.
//
// Each wave runs REPS x 64 independent instructions of one kind between two s_memtime stamps
// (shader-clock ticks) and two s_memrealtime stamps (100 MHz).  With w waves resident per SIMD, all
// running the same loop, the SIMD's cost per instruction is (ticks per wave) / (64 * REPS * w); for
// scalar instructions the compute unit's cost is that over the 4 * w waves of the unit.
// Built by x265_amd/csrc/Makefile into benchmarks/micro/issue_bench; run by bench.py.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int REPS = 256;

#define R8(body) body body body body body body body body
#define R64(body) R8(R8(body))

enum Kind { K_SAD_U8, K_PK_ADD_I16, K_MIN_I32, K_ADD_U32, K_ALIGNBYTE, K_MOV_DPP, K_S_ADD, K_S_CSELECT, K_MIX, K_QSAD, K_MQSAD,
            K_PERMLANE32_SWAP, K_MFMA_I8, K_COUNT };
static const char* kindName[K_COUNT] = { "v_sad_u8", "v_pk_add_i16", "v_min_i32", "v_add_u32", "v_alignbyte_b32",
                                          "v_mov_b32_dpp", "s_add_u32", "s_cmp+s_cselect", "valu+salu interleaved",
                                          "v_qsad_pk_u16_u8", "v_mqsad_pk_u16_u8", "v_permlane32_swap_b32", "v_mfma_i32_16x16x32_i8" };

template <int KIND>
__global__ void __launch_bounds__(1024) issue_kernel(unsigned long long* out, unsigned seed)
{
    unsigned a0 = threadIdx.x * 3u + seed, a1 = a0 ^ 0x55u, a2 = a0 + 7u, a3 = a0 * 5u, a4 = a0 + 11u, a5 = a0 ^ 0xaau, a6 = a0 + 13u,
             a7 = a0 * 9u;
    unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    // 64-bit operands of the four-offset byte SADs (a sliding 4-byte pattern against an 8-byte window, four 16-bit sums)
    unsigned long long q0 = a0 * 0x100000001ull, q1 = a1 * 0x100000003ull, q2 = a2 * 0x100000005ull, q3 = a3 * 0x100000007ull;
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    v4i_t m0 = { 0, 0, 0, 0 }, m1 = m0, m2 = m0, m3 = m0;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < REPS; rep++)
    {
        if (KIND == K_SAD_U8)
            { R8(asm volatile("v_sad_u8 %0, %0, %8, %0\n v_sad_u8 %1, %1, %8, %1\n v_sad_u8 %2, %2, %8, %2\n v_sad_u8 %3, %3, %8, %3\n"
                            "v_sad_u8 %4, %4, %8, %4\n v_sad_u8 %5, %5, %8, %5\n v_sad_u8 %6, %6, %8, %6\n v_sad_u8 %7, %7, %8, %7\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        else if (KIND == K_PK_ADD_I16)
            { R8(asm volatile("v_pk_add_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_add_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n"
                            "v_pk_add_i16 %4, %4, %8\n v_pk_add_i16 %5, %5, %8\n v_pk_add_i16 %6, %6, %8\n v_pk_add_i16 %7, %7, %8\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        else if (KIND == K_MIN_I32)
            { R8(asm volatile("v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n"
                            "v_min_i32 %4, %4, %8\n v_min_i32 %5, %5, %8\n v_min_i32 %6, %6, %8\n v_min_i32 %7, %7, %8\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        else if (KIND == K_ADD_U32)
            { R8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                            "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        else if (KIND == K_ALIGNBYTE)
            { R8(asm volatile("v_alignbyte_b32 %0, %0, %8, 1\n v_alignbyte_b32 %1, %1, %8, 1\n v_alignbyte_b32 %2, %2, %8, 1\n v_alignbyte_b32 %3, %3, %8, 1\n"
                            "v_alignbyte_b32 %4, %4, %8, 1\n v_alignbyte_b32 %5, %5, %8, 1\n v_alignbyte_b32 %6, %6, %8, 1\n v_alignbyte_b32 %7, %7, %8, 1\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        else if (KIND == K_MOV_DPP)
            { R8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                            "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                            "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                            "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        else if (KIND == K_S_ADD)
            { R8(asm volatile("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4\n"
                            "s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4\n"
                            : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(seed) : "scc");) }
        else if (KIND == K_S_CSELECT)
            { R8(asm volatile("s_cmp_lt_u32 %0, %4\n s_cselect_b32 %0, %1, %0\n s_cmp_lt_u32 %1, %4\n s_cselect_b32 %1, %2, %1\n"
                            "s_cmp_lt_u32 %2, %4\n s_cselect_b32 %2, %3, %2\n s_cmp_lt_u32 %3, %4\n s_cselect_b32 %3, %0, %3\n"
                            : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(seed) : "scc");) }
        else if (KIND == K_QSAD)
            { R8(asm volatile("v_qsad_pk_u16_u8 %0, %0, %4, %0\n v_qsad_pk_u16_u8 %1, %1, %4, %1\n v_qsad_pk_u16_u8 %2, %2, %4, %2\n v_qsad_pk_u16_u8 %3, %3, %4, %3\n"
                            "v_qsad_pk_u16_u8 %0, %0, %4, %0\n v_qsad_pk_u16_u8 %1, %1, %4, %1\n v_qsad_pk_u16_u8 %2, %2, %4, %2\n v_qsad_pk_u16_u8 %3, %3, %4, %3\n"
                            : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(seed));) }
        else if (KIND == K_MQSAD)
            { R8(asm volatile("v_mqsad_pk_u16_u8 %0, %0, %4, %0\n v_mqsad_pk_u16_u8 %1, %1, %4, %1\n v_mqsad_pk_u16_u8 %2, %2, %4, %2\n v_mqsad_pk_u16_u8 %3, %3, %4, %3\n"
                            "v_mqsad_pk_u16_u8 %0, %0, %4, %0\n v_mqsad_pk_u16_u8 %1, %1, %4, %1\n v_mqsad_pk_u16_u8 %2, %2, %4, %2\n v_mqsad_pk_u16_u8 %3, %3, %4, %3\n"
                            : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(seed));) }
        else if (KIND == K_PERMLANE32_SWAP)
            { R8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                            "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        else if (KIND == K_MFMA_I8)
            { R8(asm volatile("v_mfma_i32_16x16x32_i8 %0, %4, %5, %0\n v_mfma_i32_16x16x32_i8 %1, %4, %5, %1\n v_mfma_i32_16x16x32_i8 %2, %4, %5, %2\n v_mfma_i32_16x16x32_i8 %3, %4, %5, %3\n"
                            "v_mfma_i32_16x16x32_i8 %0, %4, %5, %0\n v_mfma_i32_16x16x32_i8 %1, %4, %5, %1\n v_mfma_i32_16x16x32_i8 %2, %4, %5, %2\n v_mfma_i32_16x16x32_i8 %3, %4, %5, %3\n"
                            : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(q0), "v"(q1));) }
        else
            { R8(asm volatile("v_sad_u8 %0, %0, %8, %0\n s_add_u32 %9, %9, %11\n v_pk_add_i16 %1, %1, %8\n s_add_u32 %10, %10, %11\n"
                            "v_min_i32 %2, %2, %8\n s_add_u32 %9, %9, %11\n v_add_u32 %3, %3, %8\n s_add_u32 %10, %10, %11\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed), "s"(s0), "s"(s1), "s"(seed) : "scc");) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3 + (unsigned)(q0 + q1 + q2 + q3) +
                          (unsigned)(m0[0] + m1[1] + m2[2] + m3[3]);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0)
    {
        out[3 * wave] = t1 - t0;
        out[3 * wave + 1] = r1 - r0;
        out[3 * wave + 2] = sink;
    }
}

template <int KIND> static int run(int wavesPerSimd, unsigned long long* dOut, std::vector<unsigned long long>& hOut, int cus,
                                   double& ticksPerWave, double& realPerWave)
{
    // wavesPerSimd 1: one 256-thread block per unit; 8: two 1024-thread blocks per unit
    const int threads = wavesPerSimd == 1 ? 256 : 1024;
    const int blocks = wavesPerSimd == 1 ? cus : cus * wavesPerSimd * 4 * 64 / threads;
    const int waves = blocks * threads / 64;
    for (int warm = 0; warm < 3; warm++)
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, dOut, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(hOut.data(), dOut, sizeof(unsigned long long) * 3 * waves, hipMemcpyDeviceToHost));
    std::vector<double> t(waves), r(waves);
    for (int i = 0; i < waves; i++)
    {
        t[i] = (double)hOut[3 * i];
        r[i] = (double)hOut[3 * i + 1];
    }
    std::nth_element(t.begin(), t.begin() + waves / 2, t.end());
    std::nth_element(r.begin(), r.begin() + waves / 2, r.end());
    ticksPerWave = t[waves / 2];
    realPerWave = r[waves / 2];
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t maxWaves = (size_t)cus * 32 + 64;
    unsigned long long* dOut = nullptr;
    CHK(hipMalloc((void**)&dOut, sizeof(unsigned long long) * 3 * maxWaves));
    std::vector<unsigned long long> hOut(3 * maxWaves);
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"instructions_per_wave\": %d, \"kinds\": {", prop.gcnArchName, cus, 64 * REPS);
    bool first = true;
    double clockSum = 0;
    int clockN = 0;
    for (int kind = 0; kind < K_COUNT; kind++)
    {
        printf("%s\"%s\": {", first ? "" : ", ", kindName[kind]);
        first = false;
        for (int w : { 1, 8 })
        {
            double ticks = 0, real = 0;
            int rc = 0;
            switch (kind)
            {
            case K_SAD_U8: rc = run<K_SAD_U8>(w, dOut, hOut, cus, ticks, real); break;
            case K_PK_ADD_I16: rc = run<K_PK_ADD_I16>(w, dOut, hOut, cus, ticks, real); break;
            case K_MIN_I32: rc = run<K_MIN_I32>(w, dOut, hOut, cus, ticks, real); break;
            case K_ADD_U32: rc = run<K_ADD_U32>(w, dOut, hOut, cus, ticks, real); break;
            case K_ALIGNBYTE: rc = run<K_ALIGNBYTE>(w, dOut, hOut, cus, ticks, real); break;
            case K_MOV_DPP: rc = run<K_MOV_DPP>(w, dOut, hOut, cus, ticks, real); break;
            case K_S_ADD: rc = run<K_S_ADD>(w, dOut, hOut, cus, ticks, real); break;
            case K_S_CSELECT: rc = run<K_S_CSELECT>(w, dOut, hOut, cus, ticks, real); break;
            case K_QSAD: rc = run<K_QSAD>(w, dOut, hOut, cus, ticks, real); break;
            case K_MQSAD: rc = run<K_MQSAD>(w, dOut, hOut, cus, ticks, real); break;
            case K_PERMLANE32_SWAP: rc = run<K_PERMLANE32_SWAP>(w, dOut, hOut, cus, ticks, real); break;
            case K_MFMA_I8: rc = run<K_MFMA_I8>(w, dOut, hOut, cus, ticks, real); break;
            default: rc = run<K_MIX>(w, dOut, hOut, cus, ticks, real); break;
            }
            if (rc)
                return rc;
            const double n = 64.0 * REPS;
            const double clockGHz = real > 0 ? ticks / real * 0.1 : 0;     // s_memrealtime ticks at 100 MHz
            if (w == 8)
            {
                clockSum += clockGHz;
                clockN++;
            }
            printf("%s\"waves_per_simd_%d\": {\"ticks_per_wave\": %.0f, \"cycles_per_instruction_per_simd\": %.3f, "
                   "\"cycles_per_instruction_per_cu\": %.3f, \"clock_GHz\": %.3f}",
                   w == 1 ? "" : ", ", w, ticks, ticks / (n * w), ticks / (n * w * 4), clockGHz);
        }
        printf("}");
    }
    printf("}, \"clock_GHz_at_8_waves\": %.3f}\n", clockN ? clockSum / clockN : 0.0);
    (void)hipFree(dOut);
    return 0;
}
