// Device-side building blocks shared by the gfx950 kernels.
//
// Lane mapping used throughout: a 64-lane wave is split into eight 8-lane
// groups.  Lane (g = lane >> 3, r = lane & 7) owns ROW r of the 8x8 block that
// group g is working on, held as packed dwords (two for 8-bit pixels, four for
// 16-bit).  One wave therefore measures up to eight candidate blocks per pass,
// which is the shape of x265's sad_x3 / sad_x4 / square-refine call sites
// (source/encoder/motion.cpp:271-353).  Row sums are combined with DPP
// (quad_perm + row_half_mirror), so no LDS is spent on reductions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xg {

constexpr int COST_MAX = 1 << 28;             // source/encoder/motion.h:65
constexpr int LOWRES_COST_MASK = (1 << 14) - 1;  // source/encoder/slicetype.h:42-43
constexpr int LOWRES_COST_SHIFT = 14;

template <typename P> struct PixT;
template <> struct PixT<uint8_t>  { static constexpr int NW = 2; static constexpr uint32_t HALF = 0x7f7f7f7fu; };
template <> struct PixT<uint16_t> { static constexpr int NW = 4; static constexpr uint32_t HALF = 0x7fff7fffu; };

// eight pixels of one row, packed little-endian
template <typename P> struct Row { uint32_t w[PixT<P>::NW]; };

// Row load from an arbitrarily aligned address.  gfx950 under HSA runs with
// unaligned global access enabled, so this lowers to one dwordx2 / dwordx4.
template <typename P> __device__ __forceinline__ Row<P> load_row(const P* p)
{
    Row<P> r;
    __builtin_memcpy(r.w, p, 8 * sizeof(P));
    return r;
}

// Same, through a pointer known to be global memory (pointers read out of a
// job descriptor are generic otherwise and would use flat loads): one
// global_load_dwordx2/x4 with a scalar base and a 32-bit lane offset.
#define XG_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));

template <typename P> __device__ __forceinline__ Row<P> load_row_g(const XG_GLOBAL P* base, int offsetPixels);
template <> __device__ __forceinline__ Row<uint8_t> load_row_g<uint8_t>(const XG_GLOBAL uint8_t* base, int offsetPixels)
{
    const u32x2_unaligned v = *(const XG_GLOBAL u32x2_unaligned*)(base + offsetPixels);
    Row<uint8_t> r;
    r.w[0] = v.x;
    r.w[1] = v.y;
    return r;
}
template <> __device__ __forceinline__ Row<uint16_t> load_row_g<uint16_t>(const XG_GLOBAL uint16_t* base, int offsetPixels)
{
    const u32x4_unaligned v = *(const XG_GLOBAL u32x4_unaligned*)(base + offsetPixels);
    Row<uint16_t> r;
    r.w[0] = v.x;
    r.w[1] = v.y;
    r.w[2] = v.z;
    r.w[3] = v.w;
    return r;
}

template <typename T> __device__ __forceinline__ const XG_GLOBAL T* as_global(const void* p)
{
    return (const XG_GLOBAL T*)p;
}

template <typename P> __device__ __forceinline__ void store_row(P* p, const Row<P>& r)
{
    __builtin_memcpy(p, r.w, 8 * sizeof(P));
}

// per-element (a + b + 1) >> 1 on packed lanes: pixelavg_pp, source/common/pixel.cpp:537-549
template <typename P> __device__ __forceinline__ Row<P> avg_row(const Row<P>& a, const Row<P>& b)
{
    Row<P> o;
#pragma unroll
    for (int i = 0; i < PixT<P>::NW; i++)
        o.w[i] = (a.w[i] | b.w[i]) - (((a.w[i] ^ b.w[i]) >> 1) & PixT<P>::HALF);
    return o;
}

template <typename P> __device__ __forceinline__ void unpack_row(const Row<P>& r, int v[8]);
template <> __device__ __forceinline__ void unpack_row<uint8_t>(const Row<uint8_t>& r, int v[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++)
        v[i] = (int)((r.w[i >> 2] >> (8 * (i & 3))) & 0xffu);
}
template <> __device__ __forceinline__ void unpack_row<uint16_t>(const Row<uint16_t>& r, int v[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++)
        v[i] = (int)((r.w[i >> 1] >> (16 * (i & 1))) & 0xffffu);
}

template <typename P> __device__ __forceinline__ Row<P> pack_row(const int v[8]);
template <> __device__ __forceinline__ Row<uint8_t> pack_row<uint8_t>(const int v[8])
{
    Row<uint8_t> r;
    r.w[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    r.w[1] = (uint32_t)v[4] | ((uint32_t)v[5] << 8) | ((uint32_t)v[6] << 16) | ((uint32_t)v[7] << 24);
    return r;
}
template <> __device__ __forceinline__ Row<uint16_t> pack_row<uint16_t>(const int v[8])
{
    Row<uint16_t> r;
#pragma unroll
    for (int i = 0; i < 4; i++)
        r.w[i] = (uint32_t)v[2 * i] | ((uint32_t)v[2 * i + 1] << 16);
    return r;
}

// ---- cross-lane moves (DPP; all 64 lanes must be active) -------------------
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7 - i inside each 8-lane half row

template <int CTRL> __device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// sum over the 8 lanes of a group, result in every lane of the group
__device__ __forceinline__ int group_sum8(int v)
{
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    return v;
}

// value held by lane 8*g, as a wave-uniform scalar
template <int G> __device__ __forceinline__ int group_value(int v)
{
    return __builtin_amdgcn_readlane(v, 8 * G);
}

// ---- LDS hand-over inside one wave ---------------------------------------------
// LDS traffic between lanes of one wave needs no hardware wait (the LDS executes
// a wave's instructions in order); this only stops the compiler from moving
// accesses across the hand-over.
__device__ __forceinline__ void lds_handover()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- block compares -----------------------------------------------------------

// row SAD on packed pixels (v_sad_u8 / v_sad_u16): source/common/pixel.cpp:40-55
template <typename P> __device__ __forceinline__ int sad_row(const Row<P>& a, const Row<P>& b);
template <> __device__ __forceinline__ int sad_row<uint8_t>(const Row<uint8_t>& a, const Row<uint8_t>& b)
{
    uint32_t s = __builtin_amdgcn_sad_u8(a.w[0], b.w[0], 0u);
    return (int)__builtin_amdgcn_sad_u8(a.w[1], b.w[1], s);
}
template <> __device__ __forceinline__ int sad_row<uint16_t>(const Row<uint16_t>& a, const Row<uint16_t>& b)
{
    uint32_t s = __builtin_amdgcn_sad_u16(a.w[0], b.w[0], 0u);
    s = __builtin_amdgcn_sad_u16(a.w[1], b.w[1], s);
    s = __builtin_amdgcn_sad_u16(a.w[2], b.w[2], s);
    return (int)__builtin_amdgcn_sad_u16(a.w[3], b.w[3], s);
}

// 8x8 SAD of the group's block; every lane of the group gets the total
template <typename P> __device__ __forceinline__ int sad8x8(const Row<P>& fenc, const Row<P>& ref)
{
    return group_sum8(sad_row<P>(fenc, ref));
}

// ---- 8x8 SATD on packed 16-bit lanes ---------------------------------------------
// 8x8 SATD = two 8x4 halves, each (sum of |4x4 Hadamard| over its two 4x4
// blocks) >> 1: source/common/pixel.cpp:239-261, 277-289.
//
// A lane holds row r of the block as four "column pairs": q[i] = (pixel i,
// pixel i + 4) in the low and high 16 bits, i.e. the same column of the left and
// the right 4x4 block, so that every butterfly of the horizontal transform is a
// packed add/sub between two registers.  The vertical transform runs across the
// four lanes of a quad (rows r ^ 1, then r ^ 2).  Its last butterfly is never
// formed: |a + b| + |a - b| = 2 max(|a|, |b|), so both lanes of a pair add
// max(|a|, |b|).  Values stay within 16 bits for 8-, 10- and 12-bit pixels
// (8 * 4095 = 32760 ahead of that last step).
typedef short s16x2 __attribute__((ext_vector_type(2)));
struct Cols { uint32_t q[4]; };

template <typename P> __device__ __forceinline__ Cols pair_cols(const Row<P>& r);
template <> __device__ __forceinline__ Cols pair_cols<uint8_t>(const Row<uint8_t>& r)
{
    Cols c;   // selector bytes: 0-3 = bytes of the second operand, 4-7 = of the first, 0x0c = zero
    c.q[0] = __builtin_amdgcn_perm(r.w[1], r.w[0], 0x0c040c00u);
    c.q[1] = __builtin_amdgcn_perm(r.w[1], r.w[0], 0x0c050c01u);
    c.q[2] = __builtin_amdgcn_perm(r.w[1], r.w[0], 0x0c060c02u);
    c.q[3] = __builtin_amdgcn_perm(r.w[1], r.w[0], 0x0c070c03u);
    return c;
}
template <> __device__ __forceinline__ Cols pair_cols<uint16_t>(const Row<uint16_t>& r)
{
    Cols c;
    c.q[0] = __builtin_amdgcn_perm(r.w[2], r.w[0], 0x05040100u);
    c.q[1] = __builtin_amdgcn_perm(r.w[2], r.w[0], 0x07060302u);
    c.q[2] = __builtin_amdgcn_perm(r.w[3], r.w[1], 0x05040100u);
    c.q[3] = __builtin_amdgcn_perm(r.w[3], r.w[1], 0x07060302u);
    return c;
}

__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// SATD of the 8x4 tile held by a quad of lanes, on packed difference columns v0..v3;
// r = the lane's row inside the tile.  `extra` (any per-lane values) is added to the
// result: it enters doubled ahead of the halving, which is exact, so that it rides on
// the same reduction.  Every lane of the quad ends up with satd_8x4 + sum(extra).
// A tile whose right half carries zero differences yields satd_4x4 of the left half.
__device__ __forceinline__ int satd8x4_cols(s16x2 v0, s16x2 v1, s16x2 v2, s16x2 v3, int r, int extra = 0)
{
    {
        const s16x2 a0 = v0 + v1, a1 = v0 - v1, a2 = v2 + v3, a3 = v2 - v3;
        v0 = a0 + a2;
        v1 = a1 + a3;
        v2 = a0 - a2;
        v3 = a1 - a3;
    }
    // rows r ^ 1: even rows form x + partner, odd rows partner - x
    const short sgn = (r & 1) ? (short)-1 : (short)1;
    const s16x2 s1 = { sgn, sgn };
    v0 = as_s16x2((uint32_t)dpp<DPP_XOR1>((int)as_u32(v0))) + s1 * v0;
    v1 = as_s16x2((uint32_t)dpp<DPP_XOR1>((int)as_u32(v1))) + s1 * v1;
    v2 = as_s16x2((uint32_t)dpp<DPP_XOR1>((int)as_u32(v2))) + s1 * v2;
    v3 = as_s16x2((uint32_t)dpp<DPP_XOR1>((int)as_u32(v3))) + s1 * v3;
    // rows r ^ 2, folded into the absolute sum
    uint32_t s = 2u * (uint32_t)extra;
    const s16x2 zero = { 0, 0 };
#define XG_SATD_LAST(V)                                                                       \
    {                                                                                         \
        const s16x2 ab = __builtin_elementwise_max(V, zero - V);                              \
        const s16x2 pb = as_s16x2((uint32_t)dpp<DPP_XOR2>((int)as_u32(ab)));                  \
        s = __builtin_amdgcn_sad_u16(as_u32(__builtin_elementwise_max(ab, pb)), 0u, s);       \
    }
    XG_SATD_LAST(v0)
    XG_SATD_LAST(v1)
    XG_SATD_LAST(v2)
    XG_SATD_LAST(v3)
#undef XG_SATD_LAST
    int t = (int)s;
    t += dpp<DPP_XOR1>(t);
    t += dpp<DPP_XOR2>(t);
    return t >> 1;
}

// 8x8 = the two 8x4 tiles of a group's two quads
__device__ __forceinline__ int satd8x8_cols(s16x2 v0, s16x2 v1, s16x2 v2, s16x2 v3, int r, int extra)
{
    int t = satd8x4_cols(v0, v1, v2, v3, r, extra);
    t += dpp<DPP_HALF_MIRROR>(t);
    return t;
}

// `f` = pair_cols of this lane's encode row, `r` its lane index inside the group (row number)
template <typename P> __device__ __forceinline__ int satd8x8(const Cols& f, const Row<P>& ref, int r, int extra = 0)
{
    const Cols d = pair_cols<P>(ref);
    return satd8x8_cols(as_s16x2(f.q[0]) - as_s16x2(d.q[0]), as_s16x2(f.q[1]) - as_s16x2(d.q[1]),
                        as_s16x2(f.q[2]) - as_s16x2(d.q[2]), as_s16x2(f.q[3]) - as_s16x2(d.q[3]), r, extra);
}

// signed 4-bit table packed in a dword, biased by 8
__device__ __forceinline__ int nib(uint32_t pack, int i) { return (int)((pack >> (4 * i)) & 15u) - 8; }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// One set of four half-pel planes (full, H, V, HV) laid out back to back, the
// way Lowres::buffer[0..3] is (source/common/lowres.cpp:160-169).
template <typename P> struct PlaneSet
{
    const P* origin;      // pixel (0,0) of plane 0
    long long planeSize;  // pixels between consecutive planes
    int stride;
    int minX, maxX, minY, maxY;  // legal block origins (inside the padded buffer)
};

// Row r of the 8x8 block at quarter-pel displacement (qx,qy) from block origin
// (bx,by): ReferencePlanes::lowresMC, source/common/lowres.h:72-93.  `anyOdd`
// is wave-uniform: when no lane has a quarter-pel component the second fetch
// and the average are skipped.  Block origins are clamped into the padded
// buffer; for vectors the reference itself would produce this is a no-op.
template <typename P>
__device__ __forceinline__ Row<P> fetch_row(const PlaneSet<P>& ps, int bx, int by, int r, int qx, int qy)
{
    const int hp = (qy & 2) | ((qx & 2) >> 1);
    const int X = clampi(bx + (qx >> 2), ps.minX, ps.maxX);
    const int Y = clampi(by + (qy >> 2), ps.minY, ps.maxY) + r;
    return load_row<P>(ps.origin + hp * ps.planeSize + (long long)Y * ps.stride + X);
}

template <typename P>
__device__ __forceinline__ Row<P> mc_row(const PlaneSet<P>& ps, int bx, int by, int r, int qx, int qy, bool anyOdd)
{
    Row<P> a = fetch_row<P>(ps, bx, by, r, qx, qy);
    if (anyOdd)
    {
        Row<P> b = fetch_row<P>(ps, bx, by, r, qx + (qx & 1), qy + (qy & 1));
        a = avg_row<P>(a, b);
    }
    return a;
}

} // namespace xg
