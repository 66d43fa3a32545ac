// Lowres motion search: one launch per anti-diagonal of the 8x8 block grid.
//
// Replaces, per block, the search half of CostEstimateGroup::estimateCUCost
// (source/encoder/slicetype.cpp:4422-4491): predictor choice by SATD over the
// already-searched neighbours, MotionEstimate::motionEstimate
// (source/encoder/motion.cpp:923-1773) on the half-pel plane set with
// ReferencePlanes::lowresQPelCost sub-pel refinement, the fork's closing
// zero-vector check and the bidir skip shortcut.
//
// Dependencies: a block reads the vectors of its right, below, below-left and
// below-right neighbours (slicetype.cpp:4438-4449).  With the grid flipped
// (x' = W-1-cuX, y' = H-1-cuY) block (x',y') depends only on blocks with a
// smaller t = x' + 2y', so all blocks of one t are independent: launch t runs
// them for every job of the batch, and the kernel boundary is the only
// synchronisation.  Occupancy comes from batching many (frame, list, distance)
// jobs per launch, never from splitting a block's serial decision chain.
//
// One wave per block.  The wave keeps the reference's candidate ORDER and
// strict-'<' tie-breaks on wave-uniform scalars, but measures the candidates
// of one step (sad_x3 / sad_x4 / square refine / four sub-pel points) in
// parallel, one per 8-lane group.
//
// The kernel is bound by vector-L1 accesses (every lane's 8-pixel row read from
// global memory is its own cache access), so the wave stages what it is about
// to re-read in its own slice of LDS with row-coalesced loads: a 32x32 window
// of the full-pel plane around the search start (re-centred if the walk leaves
// it), the two MV-cost table windows the walk prices from, and a 10-row window
// of all four half-pel planes for the sub-pel stage.  Candidate rows then come
// out of LDS as aligned dwords plus a byte-align (LDS b64/b128 reads off their
// natural alignment are replayed at 64 cycles).  Groups without a candidate
// issue no loads, and the two MV-cost reads of a candidate ride on lanes 0 and
// 1 of its group and are folded into the row reduction.
#include <cstdlib>
#include "device_common.h"
#include "kernels.h"

#define XG_LDS __attribute__((address_space(3)))

namespace xg {

// pattern tables as biased nibbles (nib()): index 0 in the low nibble
constexpr uint32_t HEX2_X = 0x679A9767u;   // hex2[8].x = -1,-2,-1,1,2,1,-1,-2   (motion.cpp:64)
constexpr uint32_t HEX2_Y = 0x8668AA86u;   // hex2[8].y = -2,0,2,2,0,-2,-2,0
constexpr uint32_t HEX6_X = 0x8879A976u;   // first six points: (-2,0)(-1,2)(1,2)(2,0)(1,-2)(-1,-2)
constexpr uint32_t HEX6_Y = 0x88668AA8u;
constexpr uint32_t SQ1_X = 0x99779788u;    // square1[1..8].x = 0,0,-1,1,-1,-1,1,1      (motion.cpp:66)
constexpr uint32_t SQ1_Y = 0x97978897u;    // square1[1..8].y = -1,1,0,0,-1,1,-1,1
constexpr int X265GPU_DIA = 0, X265GPU_HEX = 1, X265GPU_UMH = 2;   // x265gpu_config.searchMethod values
constexpr uint32_t MOD6M1 = 0x05432105u;   // mod6m1[8] = 5,0,1,2,3,4,5,0 (unbiased)          (motion.cpp:65)

template <typename P>
struct BlockCtx
{
    const XG_GLOBAL P* base;       // start of the buffer holding the 4 searched planes
    int padOffset;                 // pixels from `base` to pixel (0,0) of plane 0
    int planeSize;                 // pixels between consecutive planes
    int stride;
    int minX, maxX, minY, maxY;    // legal block origins (inside the padded buffer)
    Row<P> fr;                     // this lane's encode row, packed
    Cols f;                        // and as 16-bit column pairs for the SATD
    int bx, by;                    // block origin in pixels (wave-uniform)
    int g, r;                      // candidate group and row of this lane
    int minx, miny, maxx, maxy;    // full-pel search bounds mvmin / mvmax (wave-uniform)
    int xr, yr;                    // maxx - minx, maxy - miny
    int mvpx, mvpy;                // quarter-pel predictor the MV cost is centred on
    const XG_GLOBAL uint16_t* costBase;   // MV-cost table, entry [maxMvd] is MVD 0
    int maxMvd;
    // this wave's LDS slice and what it currently holds (all wave-uniform)
    XG_LDS uint32_t* win;          // full-pel window, WIN rows of LdsT<P>::WIN_S dwords
    XG_LDS uint32_t* sub;          // sub-pel window, 4 planes x SUB_ROWS rows of LdsT<P>::SUB_S dwords
    XG_LDS uint32_t* tab;          // MV-cost table entries of MVD -TAB_LO .. TAB_N - 1 - TAB_LO, 16 bits each (every table
                                   // reaches that far: x265gpu_set_mvcost takes none below +-1024 entries)
    bool tabAll;                   // whatever this search can price lies inside that window (wave-uniform)
#ifdef XG_STAMP
    unsigned long long st[8];
#endif
    int relX, relY;                // block origin relative to the window's first pixel
    int subRelX, subRelY;          // the same for the sub-pel window
};

#ifdef XG_STAMP
// timing build: every block of a flow launch's first job leaves the constant 100 MHz clock at the ends of its stages
// (benchmarks/chain_probe.py reads them back and walks the critical path of the dependency chain)
constexpr int XG_STAMP_BLOCKS = 1 << 16;
__device__ unsigned long long xg_stamps[XG_STAMP_BLOCKS * 8];
#define STAMP(c, i) (c).st[i] = __builtin_amdgcn_s_memrealtime()
#else
#define STAMP(c, i)
#endif

constexpr int WIN = 32;            // full-pel window side, pixels
constexpr int WIN_PAD = 12;        // pixels between the window's edge and the centre block
constexpr int SUB_ROWS = 10;       // sub-pel window: 10 rows x 16 pixels per plane
constexpr int TAB_N = 400;         // MV-cost entries held in LDS: one window around MVD 0 for both axes, staged once per
constexpr int TAB_LO = 199;        // block (+-49 pixels: what a search reaches from its predictor; beyond it, global memory)
constexpr int NOT_STAGED = 1 << 24;

// Row strides in dwords, chosen odd-ish so that the eight rows of a candidate
// fall into different banks: 12 -> banks 0,12,24,4,16,28,8,20; 20 -> 0,20,8,28,16,4,24,12;
// 4 -> 0,4,..,28; 9 -> 0,9,..,31.  Every row leaves room for the one dword the
// byte-align reads past the last pixel.
template <typename P> struct LdsT;
template <> struct LdsT<uint8_t>  { static constexpr int WIN_S = 12, SUB_S = 4; };
template <> struct LdsT<uint16_t> { static constexpr int WIN_S = 20, SUB_S = 9; };
template <typename P> struct WaveLds
{
    static constexpr int WIN_DW = WIN * LdsT<P>::WIN_S + 4;
    static constexpr int SUB_DW = 4 * SUB_ROWS * LdsT<P>::SUB_S + 4;
    static constexpr int TAB_DW = TAB_N / 2;        // 400 x 16 bit
    static constexpr int TOTAL = WIN_DW + SUB_DW + TAB_DW;
};

// sixteen pixels in flight between a global load and an LDS store
template <typename P> struct Px16 { uint32_t w[4 * sizeof(P)]; };

template <typename P> __device__ __forceinline__ Px16<P> load16_g(const XG_GLOBAL P* base, int offsetPixels)
{
    Px16<P> s;
#pragma unroll
    for (int h = 0; h < (int)sizeof(P); h++)
    {
        const u32x4_unaligned v = *(const XG_GLOBAL u32x4_unaligned*)(base + offsetPixels + h * (16 / (int)sizeof(P)));
        s.w[4 * h] = v.x;
        s.w[4 * h + 1] = v.y;
        s.w[4 * h + 2] = v.z;
        s.w[4 * h + 3] = v.w;
    }
    return s;
}

template <typename P> __device__ __forceinline__ void store16_l(XG_LDS uint32_t* dst, const Px16<P>& s)
{
#pragma unroll
    for (int i = 0; i < 4 * (int)sizeof(P); i++)
        dst[i] = s.w[i];
}

// eight pixels starting X pixels into an LDS row: aligned dword reads + byte-align
template <typename P> __device__ __forceinline__ Row<P> lds_row(const XG_LDS uint32_t* row, int X);
template <> __device__ __forceinline__ Row<uint8_t> lds_row<uint8_t>(const XG_LDS uint32_t* row, int X)
{
    const XG_LDS uint32_t* p = row + (X >> 2);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    Row<uint8_t> r;
    r.w[0] = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)X & 3u);
    r.w[1] = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)X & 3u);
    return r;
}
template <> __device__ __forceinline__ Row<uint16_t> lds_row<uint16_t>(const XG_LDS uint32_t* row, int X)
{
    const XG_LDS uint32_t* p = row + (X >> 1);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
    const uint32_t sh = ((uint32_t)X & 1u) * 2u;
    Row<uint16_t> r;
    r.w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
    r.w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
    r.w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
    r.w[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
    return r;
}

// Row r of the block displaced by the quarter-pel vector (qx,qy), from the
// half-pel plane that vector selects: ReferencePlanes::lowresMC,
// source/common/lowres.h:72-93.  Addressed the way the reference does, as one
// offset into the buffer of four planes: a predictor candidate is measured
// unclipped (slicetype.cpp:4466-4471), and one that points past the margin (the
// doubled quarter-res vector of a block that the odd-width index arithmetic of
// :4397 picked from elsewhere in the frame) lands in a neighbouring row or plane
// there too.  Only an offset outside the whole buffer, where the reference
// would leave its allocation, is pulled back.
template <typename P> __device__ __forceinline__ Row<P> fetch_plane_row(const BlockCtx<P>& c, int qx, int qy)
{
    const int hp = (qy & 2) | ((qx & 2) >> 1);
    // (rows and strides are far below 2^23: the 24-bit multiply, which issues at full rate where the 32-bit one takes
    // four slots, is exact for them; the plane offset is not multiplied that way, a plane can exceed 2^24 pixels)
    const int off = c.padOffset + hp * c.planeSize + __mul24(c.by + (qy >> 2) + c.r, c.stride) + c.bx + (qx >> 2);
    return load_row_g<P>(c.base, clampi(off, 0, 4 * c.planeSize - 8));
}

constexpr int EV_SAD = 0, EV_SATD = 1;

template <typename P> __device__ __forceinline__ Row<P> zero_row()
{
    Row<P> z;
#pragma unroll
    for (int i = 0; i < PixT<P>::NW; i++)
        z.w[i] = 0;
    return z;
}

// this lane's row of its group's candidate block, read from global memory
// (zeros when the group is idle)
template <typename P> __device__ __forceinline__ Row<P> cand_row(const BlockCtx<P>& c, int qx, int qy, bool active)
{
    Row<P> ref = zero_row<P>();
    if (active)
    {
        ref = fetch_plane_row<P>(c, qx, qy);
        if ((qx | qy) & 1)
            ref = avg_row<P>(ref, fetch_plane_row<P>(c, qx + (qx & 1), qy + (qy & 1)));
    }
    return ref;
}

// The two MV-cost table reads of (qx,qy), BitCost::mvcost (source/encoder/bitcost.h:56):
// lane 0 of the group takes the x term, lane 1 the y term, every other lane 0, so that
// summing over the group adds mvcost.
// price_mvd: the entry for the MVD `d` of lane 0 (x) / lane 1 (y) of a group out of the staged table window; an MVD outside the
// window goes to global memory, a wave-uniform branch that searches rarely take.
template <typename P> __device__ __forceinline__ int price_mvd(const BlockCtx<P>& c, int d, bool priced)
{
    const bool want = priced && c.r < 2;
    if (c.tabAll)
    {
        // (lanes that do not count may carry anything: their index is folded into the window)
        const int t = (int)((const XG_LDS uint16_t*)c.tab)[min((unsigned)(d + TAB_LO), (unsigned)(TAB_N - 1))];
        return want ? t : 0;
    }
    const bool in = (unsigned)(d + TAB_LO) < (unsigned)TAB_N;
    int t = (int)((const XG_LDS uint16_t*)c.tab)[in ? d + TAB_LO : 0];
    if (__builtin_amdgcn_ballot_w64(want && !in))
    {
        if (want && !in)
            t = (int)c.costBase[(unsigned)(clampi(d, -c.maxMvd, c.maxMvd) + c.maxMvd)];
    }
    return want ? t : 0;
}
template <typename P> __device__ __forceinline__ int price_lanes(const BlockCtx<P>& c, int qx, int qy, bool priced = true)
{
    return price_mvd<P>(c, (c.r & 1) ? qy - c.mvpy : qx - c.mvpx, priced);
}

template <typename P, int KIND> __device__ __forceinline__ int reduce_cost(const BlockCtx<P>& c, const Row<P>& ref, int t)
{
    if (KIND == EV_SAD)
        return group_sum8(sad_row<P>(c.fr, ref) + t);
    return satd8x8<P>(c.f, ref, c.r, t);
}

// cost of this lane's group candidate at quarter-pel (qx,qy), read from global
// memory; valid in every lane of an active group.  All 64 lanes call together.
template <typename P, int KIND>
__device__ __forceinline__ int eval(const BlockCtx<P>& c, int qx, int qy, bool active, bool priced)
{
    const Row<P> ref = cand_row<P>(c, qx, qy, active);
    const int t = price_lanes<P>(c, qx, qy, active && priced);
    return reduce_cost<P, KIND>(c, ref, t);
}

// mvcost of a wave-uniform vector (two scalar-address table reads)
template <typename P> __device__ __forceinline__ int mvcost_u(const BlockCtx<P>& c, int qx, int qy)
{
    const int ix = clampi(qx - c.mvpx, -c.maxMvd, c.maxMvd) + c.maxMvd;
    const int iy = clampi(qy - c.mvpy, -c.maxMvd, c.maxMvd) + c.maxMvd;
    return ((int)c.costBase[(unsigned)ix] + (int)c.costBase[(unsigned)iy]) & 0xffff;
}

template <typename P> __device__ __forceinline__ bool y_ok(const BlockCtx<P>& c, int y)
{
    return (unsigned)(y - c.miny) <= (unsigned)c.yr;
}
template <typename P> __device__ __forceinline__ bool in_range(const BlockCtx<P>& c, int x, int y)
{
    return (unsigned)(x - c.minx) <= (unsigned)c.xr && (unsigned)(y - c.miny) <= (unsigned)c.yr;
}

// ---- choosing among the candidates of one step ---------------------------------
// The reference walks the candidates of a step in a fixed order and keeps one only
// if it is strictly cheaper than the best so far (COPY2_IF_LT chains, or a packed
// (cost << k) + direction minimum, motion.cpp:263-353).  Both are the minimum of
// (cost, position-in-order) pairs with the incumbent in front, so each group
// forms the key (cost << 7 | order << 3 | group) and one DPP reduction finds the
// winner; its vector is then read from the winning group's lanes.  This keeps the
// per-candidate compare/select chains off the scalar unit.
constexpr int KEY_INVALID = 0x7fffffff;
__device__ __forceinline__ int make_key(int cost, int order, int g) { return (cost << 7) | (order << 3) | g; }

// minimum over the wave of a value that is uniform within each 8-lane group
__device__ __forceinline__ int wave_min(int v)
{
    // lanes a step does not write keep KEY_INVALID, the identity of min, which lets the compiler
    // fold each move into the v_min itself
    v = min(v, __builtin_amdgcn_update_dpp(KEY_INVALID, v, 0x128, 0xf, 0xf, false));   // row_ror:8, the row's other group
    v = min(v, __builtin_amdgcn_update_dpp(KEY_INVALID, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
    v = min(v, __builtin_amdgcn_update_dpp(KEY_INVALID, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Accept the step's winner if it beats the incumbent.  (vx,vy) are the per-lane
// candidate vectors the keys were formed for.  Returns the winner's order (0: none).
__device__ __forceinline__ int take_winner(int key, int vx, int vy, int& bx, int& by, int& bcost)
{
    if (key < (bcost << 7))
    {
        const int lane = (key & 7) << 3;
        bx = __builtin_amdgcn_readlane(vx, lane);
        by = __builtin_amdgcn_readlane(vy, lane);
        bcost = key >> 7;
        return (key >> 3) & 15;
    }
    return 0;
}

// full-pel cost of this lane's group candidate, SAD + mvcost(mv << 2), read from global memory
// (plane 0 only, one fetch: the long-range steps of the multi-hexagon search, which no window covers)
template <typename P> __device__ __forceinline__ int fpel_cost_g(const BlockCtx<P>& c, int mx, int my, bool active)
{
    Row<P> ref = zero_row<P>();
    if (active)
    {
        const int X = clampi(c.bx + mx, c.minX, c.maxX), Y = clampi(c.by + my, c.minY, c.maxY) + c.r;
        ref = load_row_g<P>(c.base, c.padOffset + __mul24(Y, c.stride) + X);
    }
    return reduce_cost<P, EV_SAD>(c, ref, price_lanes<P>(c, mx * 4, my * 4, active));
}

// The MV-cost window: 50 lanes x 8 entries; load and store halves so that callers can put other loads in flight
// between them
struct TabStage { u32x4_unaligned v; };
template <typename P> __device__ __forceinline__ TabStage tab_load(const BlockCtx<P>& c)
{
    TabStage t;
    const int lane = c.g * 8 + c.r;
    t.v = u32x4_unaligned{ 0, 0, 0, 0 };
    if (lane < TAB_N / 8)
        t.v = *(const XG_GLOBAL u32x4_unaligned*)(c.costBase + (c.maxMvd - TAB_LO + 8 * lane));
    return t;
}
template <typename P> __device__ __forceinline__ void tab_store(BlockCtx<P>& c, const TabStage& t)
{
    const int lane = c.g * 8 + c.r;
    if (lane < TAB_N / 8)
    {
        XG_LDS uint32_t* d = c.tab + 4 * lane;
        d[0] = t.v.x;
        d[1] = t.v.y;
        d[2] = t.v.z;
        d[3] = t.v.w;
    }
}

// The full-pel window around the block displaced by (mx,my): one 16-pixel load per
// lane (two lanes per window row), again split into load and store.
template <typename P> struct WinStage { Px16<P> px; int wx, wy; };
template <typename P> __device__ __forceinline__ WinStage<P> win_load(const BlockCtx<P>& c, int mx, int my)
{
    WinStage<P> w;
    const int cx = clampi(c.bx + mx, c.minX, c.maxX), cy = clampi(c.by + my, c.minY, c.maxY);
    w.wx = clampi(cx - WIN_PAD, c.minX, c.maxX + 8 - WIN);
    w.wy = clampi(cy - WIN_PAD, c.minY, c.maxY + 8 - WIN);
    const int lane = c.g * 8 + c.r, row = lane >> 1, half = lane & 1;
    w.px = load16_g<P>(c.base, c.padOffset + __mul24(w.wy + row, c.stride) + w.wx + 16 * half);
    return w;
}
template <typename P> __device__ __forceinline__ void win_store(BlockCtx<P>& c, const WinStage<P>& w)
{
    const int lane = c.g * 8 + c.r, row = lane >> 1, half = lane & 1;
    store16_l<P>(c.win + row * LdsT<P>::WIN_S + half * (4 * (int)sizeof(P)), w.px);
    c.relX = c.bx - w.wx;
    c.relY = c.by - w.wy;
}

// Stage the window around displacement (mx,my).
template <typename P> __device__ __forceinline__ void stage_search(BlockCtx<P>& c, int mx, int my)
{
    const WinStage<P> ws = win_load<P>(c, mx, my);
    lds_handover();
    win_store<P>(c, ws);
    lds_handover();
}

// Make sure every block within `ext` pixels of displacement (mx,my) lies in the staged window.
template <typename P> __device__ __forceinline__ void ensure_window(BlockCtx<P>& c, int mx, int my, int ext)
{
    const unsigned lim = (unsigned)(WIN - 8 - 2 * ext);
    if (max((unsigned)(c.relX + mx - ext), (unsigned)(c.relY + my - ext)) > lim)       // one scalar compare
        stage_search<P>(c, mx, my);
}

// this lane's row of the block displaced by full-pel (mx,my), out of the staged window
template <typename P> __device__ __forceinline__ Row<P> win_row(const BlockCtx<P>& c, int mx, int my)
{
    return lds_row<P>(c.win + __mul24(c.relY + my + c.r, LdsT<P>::WIN_S), c.relX + mx);
}

// Key of this lane's group candidate at full-pel (mx,my) out of the staged window:
// SAD + mvcost(mv << 2), invalid when the group is idle or the vector's row is
// outside mvmin.y..mvmax.y.  Idle groups pass any vector inside the window.
template <typename P>
__device__ __forceinline__ int fpel_key(const BlockCtx<P>& c, int mx, int my, bool active, int order)
{
    const Row<P> ref = win_row<P>(c, mx, my);
    const int t = price_lanes<P>(c, mx * 4, my * 4);
    const int cost = reduce_cost<P, EV_SAD>(c, ref, t);
    const bool ok = active && (unsigned)(my - c.miny) <= (unsigned)c.yr;
    return wave_min(ok ? make_key(cost, order, c.g) : KEY_INVALID);
}

// Stage rows -1..+8 and columns -1..+14 of all four half-pel planes around the
// full-pel part of quarter-pel vector (bqx,bqy), which is everything the half-
// and quarter-pel steps (up to +-3 quarter pixels from here) can read, and the
// MV-cost windows around that vector.
template <typename P> __device__ __forceinline__ void stage_subpel(BlockCtx<P>& c, int bqx, int bqy)
{
    const int sx = clampi(c.bx + (bqx >> 2) - 1, c.minX, c.maxX - 8);
    const int sy = clampi(c.by + (bqy >> 2) - 1, c.minY, c.maxY - 2);
    const int lane = c.g * 8 + c.r;
    const int plane = lane / SUB_ROWS, row = lane - plane * SUB_ROWS;
    Px16<P> px;
    if (lane < 4 * SUB_ROWS)
        px = load16_g<P>(c.base, c.padOffset + plane * c.planeSize + __mul24(sy + row, c.stride) + sx);
    lds_handover();
    if (lane < 4 * SUB_ROWS)
        store16_l<P>(c.sub + lane * LdsT<P>::SUB_S, px);
    lds_handover();
    c.subRelX = c.bx - sx;
    c.subRelY = c.by - sy;
}

template <typename P> __device__ __forceinline__ Row<P> sub_row(const BlockCtx<P>& c, int qx, int qy)
{
    const int hp = (qy & 2) | ((qx & 2) >> 1);
    return lds_row<P>(c.sub + __mul24(hp * SUB_ROWS + c.subRelY + (qy >> 2) + c.r, LdsT<P>::SUB_S), c.subRelX + (qx >> 2));
}

// cost at quarter-pel (qx,qy) out of the staged sub-pel window: ReferencePlanes::lowresQPelCost
// (source/common/lowres.h:95-140) + mvcost.  Branch-free: a vector without a quarter-pel
// part averages its block with itself.
// `anyOdd` (wave-uniform): some lane's vector has a quarter-pel part; without one the second fetch is skipped.
template <typename P, int KIND> __device__ __forceinline__ int eval_sub(const BlockCtx<P>& c, int qx, int qy, bool anyOdd)
{
    Row<P> ref = sub_row<P>(c, qx, qy);
    if (anyOdd)
        ref = avg_row<P>(ref, sub_row<P>(c, qx + (qx & 1), qy + (qy & 1)));
    return reduce_cost<P, KIND>(c, ref, price_lanes<P>(c, qx, qy));
}

// 8-point square refinement around (bmx,bmy): motion.cpp:1120-1138
template <typename P> __device__ __forceinline__ void square_refine(BlockCtx<P>& c, int& bmx, int& bmy, int& bcost)
{
    ensure_window<P>(c, bmx, bmy, 1);
    const int mx = bmx + nib(SQ1_X, c.g), my = bmy + nib(SQ1_Y, c.g);
    take_winner(fpel_key<P>(c, mx, my, true, c.g + 1), mx, my, bmx, bmy, bcost);
}

// radius-2 hexagon walk (motion.cpp:1041-1118) followed by the square refinement.
// `havePre`: the first hexagon around (bmx,bmy) was already measured together with
// the start candidates; preKey is its winner key and (preX,preY) the per-lane vectors.
template <typename P>
__device__ __forceinline__ void hex_search(BlockCtx<P>& c, int merange, int& bmx, int& bmy, int& bcost,
                                           bool havePre, int preKey, int preX, int preY)
{
    int order;
    if (havePre)
        order = take_winner(preKey, preX, preY, bmx, bmy, bcost);
    else
    {
        ensure_window<P>(c, bmx, bmy, 2);
        const int hi = c.g < 6 ? c.g : 6;      // index 6 of the tables is (0,0)
        const int mx = bmx + nib(HEX6_X, hi), my = bmy + nib(HEX6_Y, hi);
        order = take_winner(fpel_key<P>(c, mx, my, c.g < 6, c.g + 2), mx, my, bmx, bmy, bcost);
    }
    if (order)
    {
        int dir = order - 2;
        // half hexagons: three new points per step, one per group 0..2
        for (int i = (merange >> 1) - 1; i > 0 && in_range<P>(c, bmx, bmy); i--)
        {
            ensure_window<P>(c, bmx, bmy, 2);
            const int idx = dir + (c.g < 3 ? c.g : 0);
            const int mx = bmx + nib(HEX2_X, idx), my = bmy + nib(HEX2_Y, idx);
            order = take_winner(fpel_key<P>(c, mx, my, c.g < 3, c.g + 1), mx, my, bmx, bmy, bcost);
            if (!order)
                break;
            dir = (int)((MOD6M1 >> (4 * (dir + order - 1))) & 15u);
        }
    }
    square_refine<P>(c, bmx, bmy, bcost);
}

// Up to eight full-pel candidates, one per group, accepted in group order with
// the reference's strict '<' (COST_MV / COST_MV_X4 semantics, motion.cpp:263-323).
// `active` (per lane, uniform within a group) says whether the group holds a candidate that may be
// accepted; the others are neither loaded nor compared.  Callers form it from the lane's own candidate
// (a couple of vector compares) rather than as a wave-uniform bit mask over all eight candidates.
// When every candidate of the pass (block and both MV-cost entries) lies inside what is staged in
// LDS, the pass reads it there; the long-range steps, which no window covers, go to global memory.
template <typename P>
__device__ __forceinline__ void try_points(const BlockCtx<P>& c, int mx, int my, bool active, int& bmx, int& bmy,
                                           int& bcost)
{
    const bool staged = (unsigned)(c.relX + mx) <= (unsigned)(WIN - 8) && (unsigned)(c.relY + my) <= (unsigned)(WIN - 8);
    int cost;
    if (__builtin_amdgcn_ballot_w64(active && !staged) == 0)
    {
        // idle groups read the window's first block: any address inside the slice will do
        const int wx = active ? mx : -c.relX, wy = active ? my : -c.relY;
        const Row<P> ref = win_row<P>(c, wx, wy);
        cost = reduce_cost<P, EV_SAD>(c, ref, price_lanes<P>(c, 4 * mx, 4 * my, active));
    }
    else
        cost = fpel_cost_g<P>(c, mx, my, active);
    take_winner(wave_min(active ? make_key(cost, c.g + 1, c.g) : KEY_INVALID), mx, my, bmx, bmy, bcost);
}

// The horizontal arm of a long cross: its candidates all lie on the block's eight rows, two pixels apart, so the rows are
// fetched once as a strip of STRIP_W pixels (16 pixels per lane, 40 lanes) into the sub-pel window's LDS, which nothing
// uses before the sub-pel steps stage it, instead of one row per lane and candidate (the long-range steps are paid in
// vector-L1 line accesses).  `x0`: the strip's first column in the padded plane.
constexpr int STRIP_W = 80;
template <typename P> struct StripT { static constexpr int S = STRIP_W * (int)sizeof(P) / 4; };
template <typename P> __device__ __forceinline__ void stage_strip(const BlockCtx<P>& c, int x0, int oy)
{
    const int lane = c.g * 8 + c.r, row = lane / 5, chunk = lane - 5 * row;
    const int Y0 = clampi(c.by + oy, c.minY, c.maxY);
    Px16<P> px;
    if (lane < 40)
        px = load16_g<P>(c.base, c.padOffset + (Y0 + row) * c.stride + x0 + 16 * chunk);
    lds_handover();
    if (lane < 40)
        store16_l<P>(c.sub + row * StripT<P>::S + chunk * (4 * (int)sizeof(P)), px);
    lds_handover();
}
// try_points for candidates on the strip's rows
template <typename P>
__device__ __forceinline__ void try_strip_points(const BlockCtx<P>& c, int x0, int mx, int my, bool active, int& bmx, int& bmy,
                                                 int& bcost)
{
    const int col = active ? clampi(c.bx + mx, c.minX, c.maxX) - x0 : 0;
    const Row<P> ref = lds_row<P>(c.sub + c.r * StripT<P>::S, col);
    const int cost = reduce_cost<P, EV_SAD>(c, ref, price_lanes<P>(c, mx * 4, my * 4, active));
    take_winner(wave_min(active ? make_key(cost, c.g + 1, c.g) : KEY_INVALID), mx, my, bmx, bmy, bcost);
}

// COST_MV_X4 (motion.cpp:302-323) accepts each of its four points subject to the point's own row being inside
// mvmin.y..mvmax.y: for a lane that is y_ok() of its own candidate.

// CROSS(start, x_max, y_max) around (ox,oy): motion.cpp:361-385
template <typename P>
__device__ __forceinline__ void umh_cross(const BlockCtx<P>& c, int ox, int oy, int start, int x_max, int y_max,
                                          int& bmx, int& bmy, int& bcost)
{
#pragma unroll
    for (int dirY = 0; dirY < 2; dirY++)
    {
        const int lim = dirY ? y_max : x_max;
        const int room = dirY ? min(c.maxy - oy, oy - c.miny) : min(c.maxx - ox, ox - c.minx);
        int i = (short)start;
        // the horizontal arm out of a strip when it is long enough to pay for one and the strip can hold it: every
        // candidate's (clamped) block then starts within the strip's first STRIP_W - 8 columns
        const bool strip = !dirY && lim - i > 8 && 2 * lim + 8 <= STRIP_W && c.maxX + 8 - c.minX >= STRIP_W;
        int x0 = 0;
        if (strip)
        {
            x0 = clampi(c.bx + ox - lim, c.minX, c.maxX + 8 - STRIP_W);
            stage_strip<P>(c, x0, oy);
        }
        if (lim <= room)
        {
            // x4 passes (i, -i, i+2, -i-2); two reference iterations per pass
            while (i < lim - 2)
            {
                const bool two = i + 4 < lim - 2;
                const int ii = i + ((c.g >= 4) ? 4 : 0);
                const int sel = c.g & 3;
                const int d = (sel == 0) ? ii : (sel == 1) ? -ii : (sel == 2) ? ii + 2 : -ii - 2;
                const int mx = dirY ? ox : ox + d, my = dirY ? oy + d : oy;
                if (strip)
                    try_strip_points<P>(c, x0, mx, my, (c.g < 4 || two) && y_ok<P>(c, my), bmx, bmy, bcost);
                else
                    try_points<P>(c, mx, my, (c.g < 4 || two) && y_ok<P>(c, my), bmx, bmy, bcost);
                i += two ? 8 : 4;
            }
        }
        // checked single points, (+i, -i) per reference iteration; four iterations per pass
        while (i < lim)
        {
            const int ii = i + 2 * (c.g >> 1);
            const bool neg = (c.g & 1) != 0;
            const int d = neg ? -ii : ii;
            const int at = (dirY ? oy : ox) + d;       // the coordinate that moves
            const bool ok = ii < lim && (neg ? at >= (dirY ? c.miny : c.minx) : at <= (dirY ? c.maxy : c.maxx));
            if (strip)
                try_strip_points<P>(c, x0, ox + d, oy, ok, bmx, bmy, bcost);
            else
                try_points<P>(c, dirY ? ox : ox + d, dirY ? oy + d : oy, ok, bmx, bmy, bcost);
            i += 8;
        }
    }
}

constexpr uint32_t DIA1_X = 0x88889788u;   // (0,-1)(0,1)(-1,0)(1,0) then (0,0) padding
constexpr uint32_t DIA1_Y = 0x88888897u;
constexpr uint32_t OCT_X = 0x897A6978u, OCT_Y = 0xA9988776u;   // (0,-2)(-1,-1)(1,-1)(-2,0) | (2,0)(-1,1)(1,1)(0,2)
constexpr uint32_t RING_X = 0x97A6A697u, RING_Y = 0xAA997766u; // (-1,-2)(1,-2)(-2,-1)(2,-1) | (-2,1)(2,1)(-1,2)(1,2)
constexpr uint32_t CORN_X = 0x8888AA66u, CORN_Y = 0x8888A6A6u; // (-2,-2)(-2,2)(2,-2)(2,2)
// hex4[16] (motion.cpp:67-73), eight points per dword:
// x = {0,0,-2,2,-4,4,-4,4 | -4,4,-4,4,-4,4,-2,2}, y = {-4,4,-3,-3,-2,-2,-1,-1 | 0,0,1,1,2,2,3,3}
constexpr uint32_t H4X0 = 0xC4C4A688u, H4Y0 = 0x776655C4u, H4X1 = 0xA6C4C4C4u, H4Y1 = 0xBBAA9988u;

// uneven multi-hexagon search, lowres case (no extra candidates): motion.cpp:1142-1326.
// Returns true when the hexagon walk must follow.
template <typename P>
__device__ __forceinline__ bool umh_search(BlockCtx<P>& c, int pmx, int pmy, int merange, int& bmx, int& bmy,
                                           int& bcost)
{
    const int scale = 4;   // sizeScale[LUMA_8x8] = (8*8) >> 4 (motion.cpp:123-153)
#define UMH_THRESH(v) (bcost < (((v) >> 4) * scale))
    const int ucost1 = bcost;
    int cross_start = 1;
    const bool pnz = (pmx | pmy) != 0;
    {
        // DIA1 around the rounded predictor, and around (0,0) when the predictor is not zero
        const int sel = c.g & 3;
        const int mx = ((c.g < 4) ? pmx : 0) + nib(DIA1_X, sel), my = ((c.g < 4) ? pmy : 0) + nib(DIA1_Y, sel);
        try_points<P>(c, mx, my, (c.g < 4 || pnz) && y_ok<P>(c, my), bmx, bmy, bcost);
    }
    const int ucost2 = bcost;
    if ((bmx | bmy) && (bmx != pmx || bmy != pmy))
    {
        const int sel = c.g & 3;
        const int my = bmy + nib(DIA1_Y, sel);
        try_points<P>(c, bmx + nib(DIA1_X, sel), my, c.g < 4 && y_ok<P>(c, my), bmx, bmy, bcost);
    }
    if (bcost == ucost2)
        cross_start = 3;

    const int ox = bmx, oy = bmy;   // the later cross stays centred here (motion.cpp:1238)
    if (bcost == ucost2 && UMH_THRESH(2000))
    {
        {
            const int my = oy + nib(OCT_Y, c.g);
            try_points<P>(c, ox + nib(OCT_X, c.g), my, y_ok<P>(c, my), bmx, bmy, bcost);
        }
        if (bcost == ucost1 && UMH_THRESH(500))
            return false;
        if (bcost == ucost2)
        {
            const int range = (short)((merange >> 1) | 1);
            umh_cross<P>(c, ox, oy, 3, range, range, bmx, bmy, bcost);
            {
                const int my = oy + nib(RING_Y, c.g);
                try_points<P>(c, ox + nib(RING_X, c.g), my, y_ok<P>(c, my), bmx, bmy, bcost);
            }
            if (bcost == ucost2)
                return false;
            cross_start = range + 2;
        }
    }

    umh_cross<P>(c, ox, oy, cross_start, merange, merange >> 1, bmx, bmy, bcost);
    {
        const int my = oy + nib(CORN_Y, c.g & 3);
        try_points<P>(c, ox + nib(CORN_X, c.g & 3), my, c.g < 4 && y_ok<P>(c, my), bmx, bmy, bcost);
    }

    // hexagon grid of growing radius around the best so far; its first three radii (up to +-12) fit a
    // window centred here
    const int hx = bmx, hy = bmy;
    ensure_window<P>(c, hx, hy, 12);
    unsigned i = 1;
    do
    {
        const bool near = (int)(4 * i) > min(min(c.maxx - hx, hx - c.minx), min(c.maxy - hy, hy - c.miny));
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            const uint32_t tx = half ? H4X1 : H4X0;
            const uint32_t ty = half ? H4Y1 : H4Y0;
            const int dx = nib(tx, c.g);
            const int dy = nib(ty, c.g);
            const int mx = hx + dx * (int)i, my = hy + dy * (int)i;
            // near the border every point is range-checked; otherwise only the row of the
            // UNSCALED pattern offset is tested (MIN_MV, motion.cpp:1274)
            const bool ok = near ? in_range<P>(c, mx, my) : y_ok<P>(c, hy + dy);
            try_points<P>(c, mx, my, ok, bmx, bmy, bcost);
        }
    }
    while (++i <= (unsigned)(merange >> 2));
#undef UMH_THRESH
    return in_range<P>(c, bmx, bmy);
}

// radius-1 diamond: motion.cpp:1016-1039.  Order values 1, 3, 4, 12 as in the reference's
// packed direction; the winner's vector is the point itself.
template <typename P> __device__ __forceinline__ void dia_search(BlockCtx<P>& c, int merange, int& bmx, int& bmy, int& bcost)
{
    int i = merange;
    do
    {
        ensure_window<P>(c, bmx, bmy, 1);
        const int gi = c.g & 3;      // points (0,-1) (0,1) (-1,0) (1,0) = square1[1..4]
        const int mx = bmx + nib(SQ1_X, gi), my = bmy + nib(SQ1_Y, gi);
        if (!take_winner(fpel_key<P>(c, mx, my, c.g < 4, (0xC431 >> (4 * gi)) & 15), mx, my, bmx, bmy, bcost))
            break;
    }
    while (--i && in_range<P>(c, bmx, bmy));
}

// MotionEstimate::motionEstimate for a lowres reference (isLowres, no extra
// candidates, subpelRefine 1).  Returns the cost; the quarter-pel vector comes
// back in (outx, outy).  Everything returned is wave-uniform.
//
// zeroSad / zeroSatd: SAD of the searched full-pel plane and SATD of the
// closing check's plane at displacement (0,0), measured by the caller together
// with the predictor candidates so they cost no memory round trip of their own.
template <typename P, int METHOD>
__device__ __forceinline__ int motion_estimate_lowres(BlockCtx<P>& c, int mvpx, int mvpy, int merange,
                                                      int zeroSad, int zeroSatd, int mvpSad, int mv0cost, int& outx, int& outy)
{
    c.mvpx = mvpx;
    c.mvpy = mvpy;
    const int qminx = c.minx * 4, qminy = c.miny * 4, qmaxx = c.maxx * 4, qmaxy = c.maxy * 4;
    const int pmvx = clampi(mvpx, qminx, qmaxx), pmvy = clampi(mvpy, qminy, qmaxy);
    int bmx = (pmvx + 2) >> 2, bmy = (pmvy + 2) >> 2;
    const int pfx = bmx, pfy = bmy;      // predictor rounded to full pel
    {
        // How far from the predictor this search can price a vector: it starts at the clipped predictor or at zero and
        // walks at most merange pixels from there, plus the refinements; the multi-hexagon search moves its centre by
        // up to merange in the crosses, again in the hexagon grid, and then walks.  Inside the staged table window, no
        // pass needs to look.
        const int walk = METHOD == X265GPU_UMH ? 4 * (3 * merange + 8) : 4 * (merange + 4);
        const int off = max(max(abs(mvpx), abs(mvpy)), max(abs(pmvx - mvpx), abs(pmvy - mvpy)));
        c.tabAll = off + walk <= TAB_LO;
    }
    int bcost, bprecost;
    int preKey, preX, preY;
    bool hexValid = METHOD == X265GPU_HEX;
    {
        const bool sub = ((pmvx | pmvy) & 3) != 0, nz = (pmvx | pmvy) != 0;
        // the start cost is the SAD (no MV cost) of the block the clipped predictor points at:
        // the caller measured it unless clipping moved the predictor
        bprecost = mvpSad;
        if (pmvx != mvpx || pmvy != mvpy)
            bprecost = group_value<0>(sad8x8<P>(c.fr, cand_row<P>(c, pmvx, pmvy, c.g == 0)));
        // One pass out of the window (staged by the caller around a guess; re-staged here if the
        // guess was off).  Group 1: the predictor rounded to full pel (used when it has a sub-pel
        // part).  Groups 2..7: the first hexagon around it, measured speculatively: it is what the
        // hexagon search needs next unless the zero vector takes over as start point.
        ensure_window<P>(c, bmx, bmy, 2);
        const int hi = (c.g >= 2) ? c.g - 2 : 6;      // index 6 of the tables is (0,0)
        preX = bmx + nib(HEX6_X, hi);
        preY = bmy + nib(HEX6_Y, hi);
        const int cost = reduce_cost<P, EV_SAD>(c, win_row<P>(c, preX, preY), price_lanes<P>(c, preX * 4, preY * 4));
        preKey = KEY_INVALID;
        if (METHOD == X265GPU_HEX)
        {
            const bool ok = c.g >= 2 && (unsigned)(preY - c.miny) <= (unsigned)c.yr;
            preKey = wave_min(ok ? make_key(cost, c.g, c.g) : KEY_INVALID);
        }
        bcost = sub ? group_value<1>(cost) : bprecost;
        if (nz && zeroSad + mv0cost < bcost)
        {
            const int zy = max(min(0, c.maxy), c.miny);
            bcost = zeroSad + mv0cost;
            hexValid = hexValid && bmx == 0 && bmy == zy;
            bmx = 0;
            bmy = zy;
        }
    }
    if (bcost == 0)
    {
        outx = bmx * 4;
        outy = bmy * 4;
        return mvcost_u<P>(c, bmx * 4, bmy * 4);
    }

    if (METHOD == X265GPU_DIA)
        dia_search<P>(c, merange, bmx, bmy, bcost);
    else if (METHOD == X265GPU_UMH)
    {
        if (umh_search<P>(c, pfx, pfy, merange, bmx, bmy, bcost))
            hex_search<P>(c, merange, bmx, bmy, bcost, false, 0, 0, 0);
    }
    else
        hex_search<P>(c, merange, bmx, bmy, bcost, hexValid, preKey, preX, preY);

    int bqx, bqy;
    if (bprecost < bcost)
    {
        bqx = pmvx;
        bqy = pmvy;
        bcost = bprecost;
    }
    else
    {
        bqx = bmx * 4;
        bqy = bmy * 4;
    }

    // param->maxSlices > 1 would pull a full-pel result whose row left mvmin.y..mvmax.y back into range
    // here (motion.cpp:1654-1659).  Without extra candidates that cannot happen: the diamond, the hexagon
    // and every stage of the multi-hexagon search test each candidate's row (COST_MV_X4, the checked
    // single points of CROSS, the border test of the hexagon grid), and the predictor is clipped.
    if (!bcost)
        bcost = mvcost_u<P>(c, bqx, bqy);
    else
    {
        // workload[1]: four half-pel SADs, SATD at the winner, four quarter-pel
        // SATDs (motion.cpp:48-58, 1667-1699).  A candidate counts only if its own
        // row lies inside mvmin.y..mvmax.y.
        const int gi = c.g & 3;
        const unsigned qyr = (unsigned)c.yr * 4u;
        STAMP(c, 4);
        stage_subpel<P>(c, bqx, bqy);
        STAMP(c, 5);
        {
            const int qx = bqx + 2 * nib(SQ1_X, gi), qy = bqy + 2 * nib(SQ1_Y, gi);
            const int cost = eval_sub<P, EV_SAD>(c, qx, qy, ((bqx | bqy) & 1) != 0);     // half-pel steps keep the parity
            const bool ok = c.g < 4 && (unsigned)(qy - qminy) <= qyr;
            take_winner(wave_min(ok ? make_key(cost, gi + 1, c.g) : KEY_INVALID), qx, qy, bqx, bqy, bcost);
        }
        // groups 0..3: the four quarter-pel neighbours; group 4: the centre itself,
        // whose SATD replaces the running cost before the neighbours are compared
        {
            const int qx = bqx + ((c.g < 4) ? nib(SQ1_X, gi) : 0), qy = bqy + ((c.g < 4) ? nib(SQ1_Y, gi) : 0);
            const int cost = eval_sub<P, EV_SATD>(c, qx, qy, true);
            const bool ok = c.g == 4 || (c.g < 4 && (unsigned)(qy - qminy) <= qyr);
            const int key = wave_min(ok ? make_key(cost, c.g < 4 ? gi + 1 : 0, c.g) : KEY_INVALID);
            const int lane = (key & 7) << 3;
            bqx = __builtin_amdgcn_readlane(qx, lane);
            bqy = __builtin_amdgcn_readlane(qy, lane);
            bcost = key >> 7;
        }
    }

    // closing zero-vector check: SATD at (0,0) + mvcost(0,0) <= bcost replaces
    // the vector but not the cost (motion.cpp:1762-1768).  It addresses
    // fpelPlane[0] with the full lumaStride whatever plane set was searched.
    if ((bqx | bqy) && zeroSatd + mv0cost <= bcost)
        bqx = bqy = 0;
    outx = bqx;
    outy = bqy;
    return bcost;
}

// MODE_ROWS: the caller walks a block row right to left inside one launch while other
// waves walk the rows below it.  The right neighbour's vector then comes in
// (rightX,rightY) from the previous step, the vectors of the row below are read
// with L1-bypassing (sc1) loads, and this block's vector is stored write-through
// and handed back in (rightX,rightY).
// MODE_FLOW: one launch for all anti-diagonals with the blocks in diagonal order.  Every
// vector array starts out filled with FLOW_PENDING in x; a block reads its neighbours'
// vectors with sc1 loads until none of them is pending (the vector is its own ready flag)
// and stores its own write-through.  `rowCount` then points at the launch's error word.
constexpr int MODE_DIAG = 0, MODE_ROWS = 1, MODE_FLOW = 2;
constexpr int FLOW_PENDING = 0x7FFF;          // no search produces this x: vectors stay within the padded picture
constexpr int FLOW_POLL_LIMIT = 1 << 18;
template <typename P, int METHOD, int MODE, bool STAGED = false>
__device__ __forceinline__ void search_block(const SearchJob& job, const Geometry& g, int level0, int cuX, int cuY,
                                             bool lastRow, int lane, const uint16_t* costTab, int maxMvd,
                                             XG_LDS uint32_t* lds, int& rightX, int& rightY,
                                             XG_GLOBAL int* rowCount = nullptr, int done = 0)
{
    const int W = level0 ? g.W4 : g.W, H = level0 ? g.H4 : g.H;
    const int stride = level0 ? g.stride4 : g.stride;
    const int width = level0 ? g.width4 : g.width, lines = level0 ? g.lines4 : g.lines;
    const int marginX = level0 ? g.marginX4 : g.marginX, marginY = level0 ? g.marginY4 : g.marginY;
    const int padOffset = (int)(level0 ? g.padOffset4 : g.padOffset);
    const int cuXY = cuX + cuY * W;

    BlockCtx<P> c;
    c.base = as_global<P>(job.refOrigin) - padOffset;
    c.padOffset = padOffset;
    c.planeSize = (int)(level0 ? g.planeSize4 : g.planeSize);
    c.stride = stride;
    c.minX = -marginX;
    c.maxX = width + marginX - 8;
    c.minY = -marginY;
    c.maxY = lines + marginY - 8;
    c.g = lane >> 3;
    c.r = lane & 7;
    c.bx = 8 * cuX;
    c.by = 8 * cuY;
    c.costBase = as_global<uint16_t>(costTab) - maxMvd;
    c.maxMvd = maxMvd;
    c.mvpx = c.mvpy = 0;
    c.win = lds;
    c.sub = lds + WaveLds<P>::WIN_DW;
    c.tab = lds + WaveLds<P>::WIN_DW + WaveLds<P>::SUB_DW;
#ifdef XG_POISON_LDS
    // test build: whatever the wave reads from LDS without having staged it first shows up
    // as a result that changes with this constant
    for (int i = lane; i < WaveLds<P>::TOTAL; i += 64)
        lds[i] = XG_POISON_LDS;
    lds_handover();
#endif
    c.relX = c.relY = c.subRelX = c.subRelY = NOT_STAGED;
    c.tabAll = false;
    STAMP(c, 0);
    STAMP(c, 7);
    // search bounds that stay inside the extended frame (slicetype.cpp:4417-4420)
    c.minx = -cuX * 8 - 8;
    c.miny = -cuY * 8 - 8;
    c.maxx = (W - cuX - 1) * 8 + 8;
    c.maxy = (H - cuY - 1) * 8 + 8;
    c.xr = c.maxx - c.minx;
    c.yr = c.maxy - c.miny;
    // group 0 reads the encode block; the other groups take their row from it across lanes, but only once the
    // neighbour vectors and everything they lead to have been put in flight behind this load: the first use of the
    // rows is the predictor SATD, and one memory round trip carries both
    if (c.g == 0)
        c.fr = load_row_g<P>(as_global<P>(job.fencPlane0) - padOffset, padOffset + (c.by + c.r) * stride + c.bx);

    // ---- predictor: lowest-SATD neighbour vector (slicetype.cpp:4438-4476) ----
    const XG_GLOBAL int32_t* mvs = as_global<int32_t>(job.mvs);
    const bool v0 = cuX < W - 1, v1 = !lastRow, v2 = !lastRow && cuX > 0, v3 = !lastRow && cuX < W - 1;
    bool v4 = false;
    int hx = 0, hy = 0;
    if (job.hmeMvs && !level0)
    {
        const int cuXY4 = (cuX / 2) + (cuY / 2) * W / 2;
        if (as_global<int32_t>(job.hmeMvCosts)[cuXY4] > 0)
        {
            v4 = true;
            hx = as_global<int32_t>(job.hmeMvs)[2 * cuXY4] * 2;
            hy = as_global<int32_t>(job.hmeMvs)[2 * cuXY4 + 1] * 2;
        }
    }
    int mvpx = 0, mvpy = 0, skipCost = 0x7fffffff;
    int zeroSad, zeroSatd, mvpSad, mv0cost;
    int mx = 0, my = 0, fencCost = 0;
    {
        // one pass: groups 0..4 = the predictor candidates (SATD); group 5 = the
        // block at displacement (0,0) in the plane the closing zero-vector check
        // reads (SATD); group 6 = (0,0) in the searched full-pel plane (SAD)
        // STAGED, bottom row: a block there has one neighbour, and nothing to assume its vector from.  Group 1, idle
        // in that row, stands in: it takes the vector of the block two to the right, which the chain produces one
        // step before the neighbour's, and measures it like a candidate's; the predictor choice never looks at it.
        const bool standIn = STAGED && lastRow;
        const bool m1 = standIn ? cuX < W - 2 : v1;
        const int off = (c.g == 0) ? 1 : (c.g == 1) ? (standIn ? 2 : W) : (c.g == 2) ? W - 1 : W + 1;
        const bool mine = (c.g == 0) ? v0 : (c.g == 1) ? m1 : (c.g == 2) ? v2 : (c.g == 3) ? v3 : false;
        // STAGED (MODE_FLOW launches of few searches, which are paid in the length of the dependent chain): the
        // neighbour vectors are taken as they come.  The block below right finished three steps of the chain ago,
        // the block below two, and the blocks to the right and below left are the two this block is handed out
        // behind: they finish last.  Whatever hangs only on the vectors that are there at the first look (their
        // candidate blocks, both zero blocks, the window around the first guess, the table window, their SATDs) is
        // done at once; the others are then waited for in the order they come in, and one that equals a vector
        // already measured (a smooth field's usual case) costs no memory round trip between its arrival and the
        // search.  One that does not is fetched and measured in a pass of its own.  A launch with more searches
        // than the chip holds waves is paid per instruction instead, and takes the vectors in one wait.
        int cx = 0, cy = 0;
        long long flowV = FLOW_PENDING;
        // wait for the vectors of the lanes that `want` one (the vector is its own ready flag)
        auto flow_wait = [&](bool want, bool nearby)
        {
            int polls = 0;
            for (;;)
            {
                if (want && (!STAGED || (int)flowV == FLOW_PENDING))
                    flowV = __hip_atomic_load((const XG_GLOBAL long long*)(mvs + 2 * (cuXY + off)), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
                if (!__builtin_amdgcn_ballot_w64(want && (int)flowV == FLOW_PENDING))
                    break;
                // never expected to run out (a block only waits for blocks handed out before it); rather than hang
                // the device, stop waiting here and, through the error word, everywhere, and let the host call fail
                if ((++polls & 255) == 0)
                {
                    int err = __hip_atomic_load(rowCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    err = __builtin_amdgcn_readfirstlane(err);
                    if (err || polls > FLOW_POLL_LIMIT)
                    {
                        if (lane == 0)
                            __hip_atomic_store(rowCount, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                // a vector one step of the chain away is looked for often; one further off, which is what a wave
                // handed out far ahead of the wavefront waits for first, less and less often: every look is a
                // request to the L2 channel that the wavefront's own hand-overs go through
                if (!STAGED)
                    __builtin_amdgcn_s_sleep(8);
                else if (nearby)
                    __builtin_amdgcn_s_sleep(2);
                else if (polls < 16)
                    __builtin_amdgcn_s_sleep(8);
                else
                    __builtin_amdgcn_s_sleep(32);
            }
            // a wait that was given up leaves the pending mark behind: the zero vector stands in for it (the host
            // discards these results and runs the batch again)
            if (want && (int)flowV == FLOW_PENDING)
                flowV = 0;
        };
        bool later = false;                     // this lane's vector was still pending at the first look
        if (MODE == MODE_FLOW)
        {
            if (STAGED)
            {
                if (mine)
                    flowV = __hip_atomic_load((const XG_GLOBAL long long*)(mvs + 2 * (cuXY + off)), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
                later = mine && (int)flowV == FLOW_PENDING;
            }
            else
                flow_wait(mine, false);
            if (mine && !later)
            {
                cx = (int)flowV;
                cy = (int)(flowV >> 32);
            }
        }
        else if (MODE == MODE_ROWS)
        {
            if (mine && c.g != 0)
            {
                const long long v = __hip_atomic_load((const XG_GLOBAL long long*)(mvs + 2 * (cuXY + off)), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
                cx = (int)v;
                cy = (int)(v >> 32);
            }
            if (c.g == 0)
            {
                cx = rightX;     // unused (v0 false) at the first block of a row
                cy = rightY;
            }
        }
        else if (mine)
        {
            const u32x2_unaligned v = *(const XG_GLOBAL u32x2_unaligned*)(mvs + 2 * (cuXY + off));
            cx = (int)v.x;
            cy = (int)v.y;
        }
        if (c.g == 4)
        {
            cx = hx;
            cy = hy;
        }
        // bit 8 G: group G's vector is still to come (wave-uniform)
        const unsigned long long laterMask = STAGED ? __builtin_amdgcn_ballot_w64(later) : 0ull;
#define AT_HAND(G) (!STAGED || !((laterMask >> (8 * G)) & 1))
        // A guess at where the search will start, so that its window can be in flight together
        // with the candidate blocks: the first neighbour vector at hand, treated the way the predictor
        // will be.  The MV-cost windows are staged around MVD 0, where a search near its
        // predictor prices from.
        int sx = 0, sy = 0;
        if (v0 && AT_HAND(0)) { sx = group_value<0>(cx); sy = group_value<0>(cy); }
        else if (m1 && AT_HAND(1)) { sx = group_value<1>(cx); sy = group_value<1>(cy); }
        else if (STAGED && v3 && AT_HAND(3)) { sx = group_value<3>(cx); sy = group_value<3>(cy); }
        else if (v4) { sx = group_value<4>(cx); sy = group_value<4>(cy); }
        sx = (clampi(sx, c.minx * 4, c.maxx * 4) + 2) >> 2;
        sy = (clampi(sy, c.miny * 4, c.maxy * 4) + 2) >> 2;

        const bool cand = (mine && !later) || (c.g == 4 && v4);
        // The kernel pays for every line a load touches (the vector L1 serves about one line per two cycles per
        // compute unit whether it hits or not), so nothing is fetched twice.  Neighbouring blocks mostly carry the
        // same vector: a candidate equal to one of another group takes that group's figures instead of reading the
        // same eight rows again (equal vectors have equal costs, so which of them the strict '<' chain below picks
        // makes no difference to what it returns).  The zero block of
        // the searched plane is the zero block of the closing check whenever both are plane 0 of the same picture.
        // The table entry of mvcost(0,0) comes out of the table window staged below when it lies inside it.
        // (both components fit 16 bits: vectors stay within the padded picture, in quarter pixels)
        int origin = c.g;
        {
            const int vec = (cx & 0xffff) | (int)((unsigned)cy << 16);
            // branch-free, highest group first, so that the lowest group with the same vector is the one that stays
#define SAME_AS(G, VALID)                                                                    \
            {                                                                                \
                const bool same = (VALID) & cand & (c.g > G) & (vec == group_value<G>(vec)); \
                origin = same ? G : origin;                                                  \
            }
            SAME_AS(3, v3 && AT_HAND(3))
            SAME_AS(2, v2 && AT_HAND(2))
            SAME_AS(1, m1 && AT_HAND(1))
            SAME_AS(0, v0 && AT_HAND(0))
#undef SAME_AS
        }
#undef AT_HAND
        const bool dup = origin != c.g;
        const bool zeroShared = job.zeroRef == job.refOrigin && job.zeroStride == stride;     // wave-uniform
        Row<P> ref = cand_row<P>(c, cx, cy, (cand && !dup) || (c.g == 6 && !zeroShared));
        if (c.g == 5)
            ref = load_row_g<P>(as_global<P>(job.zeroRef), c.by * stride + c.bx + c.r * job.zeroStride);
        const WinStage<P> ws = win_load<P>(c, sx, sy);
        const TabStage ts = tab_load<P>(c);
        lds_handover();
        win_store<P>(c, ws);
        tab_store<P>(c, ts);
        lds_handover();
        // table entries of mvcost(0,0) should this group's vector become the predictor
        // (group 6 stands in for the zero predictor of a block without neighbours)
        int t0 = price_mvd<P>(c, c.r == 0 ? -cx : -cy, (cand || c.g == 6) && !dup);
#pragma unroll
        for (int i = 0; i < PixT<P>::NW; i++)
            c.fr.w[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * c.r, (int)c.fr.w[i]);
        c.f = pair_cols<P>(c.fr);
        if (zeroShared)
        {
#pragma unroll
            for (int i = 0; i < PixT<P>::NW; i++)
            {
                const uint32_t fromG5 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (40 + c.r), (int)ref.w[i]);
                ref.w[i] = c.g == 6 ? fromG5 : ref.w[i];
            }
        }

        int cost = reduce_cost<P, EV_SATD>(c, ref, 0);
        int sad = sad8x8<P>(c.fr, ref);
        t0 += dpp<DPP_XOR1>(t0);
        STAMP(c, 1);
        STAMP(c, 2);
        // ---- STAGED: the vectors that were still to come, in the order the chain produces them ----
        // bit 8 G: group G holds figures it measured itself
        unsigned long long measured = STAGED ? __builtin_amdgcn_ballot_w64(cand && !dup) : 0ull;
        const int turn = c.g == 3 ? 0 : c.g == 1 ? 1 : 2;
        auto take_turn = [&](int k)
        {
            const bool want = later && turn == k;
            if (!__builtin_amdgcn_ballot_w64(want))
                return;
            flow_wait(want, k > 0);
            STAMP(c, 2);
            if (want)
            {
                cx = (int)flowV;
                cy = (int)(flowV >> 32);
            }
            {
                // the lowest group that measured the same vector (a waiting group has measured nothing yet, so
                // it never finds itself); of the last turn's two, which may carry the same new vector, the
                // lower group fetches it
                const int vec = (cx & 0xffff) | (int)((unsigned)cy << 16);
                int found = c.g;
#define SAME_AS(G)                                                                            \
                {                                                                             \
                    const bool same = (((measured >> (8 * G)) & 1) != 0) & (vec == group_value<G>(vec)); \
                    found = same ? G : found;                                                 \
                }
                SAME_AS(4)
                SAME_AS(3)
                SAME_AS(2)
                SAME_AS(1)
                SAME_AS(0)
#undef SAME_AS
                const bool g0waits = (__builtin_amdgcn_ballot_w64(want) & 1) != 0;
                const bool g0new = g0waits & (__builtin_amdgcn_readlane(found, 0) == 0);
                const bool asG0 = g0new & (c.g == 2) & (found == 2) & (vec == group_value<0>(vec));
                found = asG0 ? 0 : found;
                origin = want ? found : origin;
            }
            const bool fetch = want && origin == c.g;
            const unsigned long long fetching = __builtin_amdgcn_ballot_w64(fetch);
            if (fetching)
            {
                // a vector that no group has measured yet: its block, and a pass for it
                const Row<P> ref2 = cand_row<P>(c, cx, cy, fetch);
                int t0b = price_mvd<P>(c, c.r == 0 ? -cx : -cy, fetch);
                const int cost2 = reduce_cost<P, EV_SATD>(c, ref2, 0);
                const int sad2 = sad8x8<P>(c.fr, ref2);
                t0b += dpp<DPP_XOR1>(t0b);
                cost = fetch ? cost2 : cost;
                sad = fetch ? sad2 : sad;
                t0 = fetch ? t0b : t0;
                measured |= fetching;
            }
        };
        // While vectors are still out, the search runs on an assumption: that they will all equal the neighbour
        // vector the chain produced last (below right, then below).  The vectors are then looked at turn by turn as
        // they come: while they agree the result stands, and once the last turn's have agreed it is the block's
        // result, since every figure that goes into it is a function of the vectors alone; a step of the chain is
        // then a hand-over and a compare.  A vector that disagrees is taken the long way, and the search runs again
        // on what is known by then (the vectors that are still out assumed as before).  The bottom row, which has
        // only its right neighbour, waits.
        const bool mayAssume = STAGED && m1;                                   // wave-uniform
        int nextTurn = 0;                                                      // the turns before it have been taken
        if (STAGED && laterMask)
        {
            const int upTo = mayAssume ? (v3 ? 1 : 2) : 3;
#pragma unroll 1
            for (; nextTurn < upTo; nextTurn++)
                take_turn(nextTurn);
        }
        bool searched = false;                 // a search has run, from the start values in `was`
        int was[7] = { 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll 1
        for (;;)
        {
            const bool assumed = STAGED && later && turn >= nextTurn;
            const bool anyAssumed = mayAssume && __builtin_amdgcn_ballot_w64(assumed) != 0;
            int ux = cx, uy = cy, uorigin = origin;
            if (STAGED && anyAssumed)
            {
                const int from = (nextTurn >= 2 || !v3) ? 8 : 24;              // first lane of the group assumed from
                const int ax = __builtin_amdgcn_readlane(cx, from), ay = __builtin_amdgcn_readlane(cy, from);
                const int ao = __builtin_amdgcn_readlane(origin, from);
                ux = assumed ? ax : cx;
                uy = assumed ? ay : cy;
                uorigin = assumed ? ao : origin;
            }
            // groups whose vector another group measured take that group's figures (without STAGED such a group is
            // always the higher one, which the strict '<' below never picks: only its cost is looked at)
            const bool copy = uorigin != c.g;
            const int costO = __builtin_amdgcn_ds_bpermute(32 * uorigin, cost);
            const int ucost = copy ? costO : cost;
            int usad = sad, ut0 = t0;
            if (STAGED)
            {
                const int sadO = __builtin_amdgcn_ds_bpermute(32 * uorigin, sad);
                const int t0O = __builtin_amdgcn_ds_bpermute(32 * uorigin, t0);
                usad = copy ? sadO : sad;
                ut0 = copy ? t0O : t0;
            }
            zeroSatd = group_value<5>(ucost);
            zeroSad = group_value<6>(usad);
            int mvpcost = COST_MAX;
            int wg = 6;
            mvpx = mvpy = 0;
            skipCost = 0x7fffffff;
#define TAKE(G, VALID)                                                                      \
            if (VALID)                                                                      \
            {                                                                               \
                const int cg = group_value<G>(ucost);                                       \
                if (cg < mvpcost)                                                           \
                {                                                                           \
                    mvpcost = cg;                                                           \
                    mvpx = group_value<G>(ux);                                              \
                    mvpy = group_value<G>(uy);                                              \
                    wg = G;                                                                 \
                }                                                                           \
                /* tracks the latest candidate while the running best is the zero vector */\
                if (!(mvpx | mvpy) && job.bBidir)                                           \
                    skipCost = cg;                                                          \
            }
            TAKE(0, v0)
            TAKE(1, v1)
            TAKE(2, v2)
            TAKE(3, v3)
            TAKE(4, v4)
#undef TAKE
            mvpSad = __builtin_amdgcn_readlane(usad, wg * 8);
            mv0cost = __builtin_amdgcn_readlane(ut0, wg * 8);

            STAMP(c, 3);
            // (a vector that disagreed with the assumption often loses to the assumed one all the same: the search
            // then starts from what it started from before, and its result stands)
            const bool sameStart = STAGED && searched && mvpx == was[0] && mvpy == was[1] && mvpSad == was[2] &&
                                   mv0cost == was[3] && zeroSad == was[4] && zeroSatd == was[5] && skipCost == was[6];
            if (!sameStart)
            {
                fencCost = motion_estimate_lowres<P, METHOD>(c, mvpx, mvpy, job.merange, zeroSad, zeroSatd, mvpSad, mv0cost, mx, my);
                c.tabAll = false;           // what is priced outside a search (a late vector's zero entry) looks again
                // bidir skip shortcut (slicetype.cpp:4485-4489)
                if (skipCost < 64 && skipCost < fencCost && job.bBidir)
                {
                    fencCost = skipCost;
                    mx = my = 0;
                }
            }
            if (!STAGED || !anyAssumed)
                break;
            searched = true;
            was[0] = mvpx;
            was[1] = mvpy;
            was[2] = mvpSad;
            was[3] = mv0cost;
            was[4] = zeroSad;
            was[5] = zeroSatd;
            was[6] = skipCost;
            // the assumption's test, turn by turn
            bool agreed = true;
#pragma unroll 1
            for (int k = nextTurn; k < 3; k++)
            {
                const bool lanes = assumed && turn == k;
                if (!__builtin_amdgcn_ballot_w64(lanes))
                    continue;
                flow_wait(lanes, true);
                if (__builtin_amdgcn_ballot_w64(lanes && ((int)flowV != ux || (int)(flowV >> 32) != uy)))
                {
                    agreed = false;
#pragma unroll 1
                    for (; nextTurn <= k; nextTurn++)
                        take_turn(nextTurn);
                    break;
                }
            }
            if (agreed)
            {
                STAMP(c, 2);
                STAMP(c, 3);
                STAMP(c, 4);
                STAMP(c, 5);
                break;
            }
        }
    }

    if (MODE == MODE_ROWS)
    {
        // Announce the `done` blocks before this one: the previous block's vector was stored a
        // whole search ago, and this drain makes sure of it without waiting on anything new.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(rowCount, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STAMP(c, 6);
    if (lane == 0)
    {
        if (MODE != MODE_DIAG)
            __hip_atomic_store((XG_GLOBAL long long*)(as_global<int32_t>(job.mvs) + 2 * cuXY),
                               (long long)(((unsigned long long)(unsigned)my << 32) | (unsigned)mx), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        else
        {
            job.mvs[2 * cuXY] = mx;
            job.mvs[2 * cuXY + 1] = my;
        }
        job.mvCosts[cuXY] = fencCost;
    }
    rightX = mx;
    rightY = my;
#ifdef XG_STAMP
    if (MODE == MODE_FLOW && blockIdx.x == 0 && cuXY < XG_STAMP_BLOCKS && lane < 8)
        xg_stamps[cuXY * 8 + lane] = c.st[lane];
#endif
}

#ifdef XG_STAMP
extern "C" int x265gpu_debug_read_stamps(void* out, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xg_stamps), bytes < sizeof(xg_stamps) ? bytes : sizeof(xg_stamps));
}
#endif

// Block rows [first, last] of cooperative slice `slice` (slicetype.cpp:4265-4266); one slice: the frame.
__device__ __forceinline__ void slice_rows(const SearchJob& job, int H, int slice, int& first, int& last)
{
    first = job.sliceRows * slice;
    last = (slice == job.numSlices - 1) ? H - 1 : job.sliceRows * (slice + 1) - 1;
}

// The block row starts a slice's bottom-up walk: no predictors from the row below (lastRow, slicetype.cpp:4267).
__device__ __forceinline__ bool slice_bottom_row(const SearchJob& job, int H, int cuY)
{
    if (cuY == H - 1)
        return true;
    const int below = cuY + 1;
    return job.numSlices > 1 && below % job.sliceRows == 0 && below / job.sliceRows < job.numSlices;
}

// grid: (ceil(blocksOnDiagonal / 4), jobs, slices); 4 waves per workgroup, one block each.
// Every cooperative slice is a wavefront of its own, counted from the slice's bottom row.
// The search method is a compile-time parameter: every job of a launch shares it,
// and the hexagon-only kernel stays free of the multi-hexagon code's registers.
template <typename P, int METHOD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) search_diag_kernel(const SearchJob* jobs, Geometry g, int level0, int t,
                                                          const uint16_t* costTab, int maxMvd)
{
    __shared__ uint32_t ldsBuf[4 * WaveLds<P>::TOTAL];
    const int W = level0 ? g.W4 : g.W, H = level0 ? g.H4 : g.H;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction
    const int k = blockIdx.x * 4 + wave;
    const SearchJob job = jobs[blockIdx.y];
    const int slice = blockIdx.z;
    if (slice >= job.numSlices)
        return;
    int first, last;
    slice_rows(job, H, slice, first, last);
    const int ylo = max(0, (t - (W - 1) + 1) >> 1), yhi = min(last - first, t >> 1);
    const int yq = ylo + k;
    if (yq > yhi)
        return;                       // whole wave leaves together
    const int xq = t - 2 * yq;
    int rx = 0, ry = 0;
    search_block<P, METHOD, MODE_DIAG>(job, g, level0, W - 1 - xq, last - yq, yq == 0, threadIdx.x & 63, costTab, maxMvd,
                                   (XG_LDS uint32_t*)ldsBuf + wave * WaveLds<P>::TOTAL, rx, ry);
}

// ---- one launch for all anti-diagonals ------------------------------------------
// Work item = one block row of one job; items are handed out in the order (row from
// the bottom, job), so whoever holds an item only ever waits for an item that was
// handed out before it, to a wave that is running: no launch-order or residency
// assumption is needed for progress.  A wave walks its row right to left; before
// block x' it needs the row below to have finished x' + 2 blocks (below-left is the
// last of the three vectors it reads there).  Every row publishes its count of
// finished blocks in sync[SYNC_ROWS + job * H + row]; a block is counted one search
// later, when its vector's write-through store has certainly drained, so that the
// publishing wave never sits waiting for that store.
constexpr int SYNC_HEAD = 0, SYNC_ERROR = 64, SYNC_ROWS = 128;   // each on 256-byte lines of its own
constexpr int POLL_LIMIT = 1 << 20;

template <typename P, int METHOD>
__device__ __forceinline__ void search_rows_body(const SearchJob* jobs, int nJobs, const Geometry& g, int level0,
                                                 const uint16_t* costTab, int maxMvd, int* sync, XG_LDS uint32_t* lds)
{
    const int W = level0 ? g.W4 : g.W, H = level0 ? g.H4 : g.H;
    const int lane = threadIdx.x & 63;
    XG_GLOBAL int* gsync = (XG_GLOBAL int*)sync;
    const int nItems = nJobs * H;
    bool gaveUp = false;
    for (;;)
    {
        int item = 0;
        if (lane == 0)
            item = __hip_atomic_fetch_add(gsync + SYNC_HEAD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nItems)
            break;
        const int yq = item / nJobs, jobIdx = item - yq * nJobs;
        const SearchJob job = jobs[jobIdx];
        XG_GLOBAL int* prog = gsync + SYNC_ROWS + jobIdx * H + yq;     // this row's count; the row below is prog[-1]
        const bool bottom = slice_bottom_row(job, H, H - 1 - yq);
        const bool waits = !bottom;
        int seen = 0, ahead = 0, rx = 0, ry = 0;
        for (int xq = 0; xq < W; xq++)
        {
            if (waits && !gaveUp)
            {
                const int need = min(xq + 2, W);
                seen = max(seen, __builtin_amdgcn_readfirstlane(ahead));     // the read put in flight one block ago
                int polls = 0;
                while (seen < need)
                {
                    seen = __hip_atomic_load(prog - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    seen = __builtin_amdgcn_readfirstlane(seen);
                    if (seen >= need)
                        break;
                    // never expected: rather than hang the device, stop waiting (here and, through the
                    // error word, in every other wave) and report it; the host discards the results
                    if ((++polls & 1023) == 0)
                    {
                        int err = __hip_atomic_load(gsync + SYNC_ERROR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        err = __builtin_amdgcn_readfirstlane(err);
                        if (err || polls > POLL_LIMIT)
                        {
                            if (lane == 0)
                                __hip_atomic_store(gsync + SYNC_ERROR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            gaveUp = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                asm volatile("" ::: "memory");
                // look at the row below's count again while this block is searched
                if (seen < W)
                    ahead = __hip_atomic_load(prog - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            search_block<P, METHOD, MODE_ROWS>(job, g, level0, W - 1 - xq, H - 1 - yq, bottom, lane, costTab, maxMvd, lds, rx, ry,
                                          prog, xq);
        }
        // the last vector's store must have left before the count that announces it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(prog, W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// 8 workgroups per compute unit need at most 80 SGPRs and 64 VGPRs
template <typename P, int METHOD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8)))
search_rows_kernel(const SearchJob* jobs, int nJobs, Geometry g, int level0, const uint16_t* costTab, int maxMvd, int* sync)
{
    __shared__ uint32_t ldsBuf[4 * WaveLds<P>::TOTAL];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    search_rows_body<P, METHOD>(jobs, nJobs, g, level0, costTab, maxMvd, sync,
                                (XG_LDS uint32_t*)ldsBuf + wave * WaveLds<P>::TOTAL);
}

// the multi-hexagon search does not fit that register budget without spilling to scratch
template <typename P, int METHOD>
__global__ void __launch_bounds__(256)
search_rows_kernel_wide(const SearchJob* jobs, int nJobs, Geometry g, int level0, const uint16_t* costTab, int maxMvd, int* sync)
{
    __shared__ uint32_t ldsBuf[4 * WaveLds<P>::TOTAL];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    search_rows_body<P, METHOD>(jobs, nJobs, g, level0, costTab, maxMvd, sync,
                                (XG_LDS uint32_t*)ldsBuf + wave * WaveLds<P>::TOTAL);
}

// ---- one launch for all anti-diagonals, vectors as ready flags -------------------
// grid: (jobs, chunks); chunk c = up to four consecutive blocks of one anti-diagonal, chunks sorted by diagonal
// (chunkTab[c] = t | first block << 16).  Workgroups are handed out in the order of their linear index, x (the
// job) fastest: a block only ever waits for blocks of lower chunks, i.e. of workgroups handed out before its own,
// so the lowest unfinished workgroup can always run.  Should the hardware not hold to that, the bounded poll in
// search_block ends the wait and the call fails.
// (the STAGED form serves launches that are paid in the length of their dependent chain, not in resident waves: it
// gets the registers its second look at the block needs)
template <typename P, int METHOD, bool STAGED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(STAGED ? 4 : 8)))
search_flow_kernel(const SearchJob* jobs, int nJobs, Geometry g, int level0, const uint32_t* chunkTab, const uint16_t* costTab,
                   int maxMvd, int* sync)
{
    __shared__ uint32_t ldsBuf[4 * WaveLds<P>::TOTAL];
    if ((int)blockIdx.x >= nJobs)
        return;
    const int W = level0 ? g.W4 : g.W, H = level0 ? g.H4 : g.H;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t chunk = chunkTab[blockIdx.y];
    const int t = (int)(chunk & 0xffffu), k = (int)(chunk >> 16) + wave;
    const SearchJob job = jobs[blockIdx.x];
    const int ylo = max(0, (t - (W - 1) + 1) >> 1), yhi = min(H - 1, t >> 1);
    const int yq = ylo + k;
    if (yq > yhi)
        return;
    const int xq = t - 2 * yq;
    int rx = 0, ry = 0;
    search_block<P, METHOD, MODE_FLOW, STAGED>(job, g, level0, W - 1 - xq, H - 1 - yq, yq == 0, threadIdx.x & 63, costTab, maxMvd,
                                       (XG_LDS uint32_t*)ldsBuf + wave * WaveLds<P>::TOTAL, rx, ry,
                                       (XG_GLOBAL int*)sync + SYNC_ERROR, 0);
}

// every vector of the jobs' arrays pending
__global__ void __launch_bounds__(256) search_flow_fill_kernel(const SearchJob* jobs, int blocks)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < blocks)
        jobs[blockIdx.y].mvs[2 * i] = FLOW_PENDING;
}

int search_flow_chunks(int W, int H, uint32_t* out)
{
    int n = 0;
    for (int t = 0; t <= (W - 1) + 2 * (H - 1); t++)
    {
        const int ylo = (t - (W - 1) + 1) >> 1 > 0 ? (t - (W - 1) + 1) >> 1 : 0;
        const int yhi = (t >> 1) < H - 1 ? (t >> 1) : H - 1;
        for (int k = 0; k <= yhi - ylo; k += 4, n++)
            if (out)
                out[n] = (uint32_t)t | ((uint32_t)k << 16);
    }
    return n;
}

template <typename P, bool STAGED>
static void launch_flow_method(int method, hipStream_t s, dim3 grid, const SearchJob* dJobs, int n, const Geometry& g,
                               int level0, const uint32_t* chunkTab, const uint16_t* costTab, int maxMvd, int* sync)
{
    if (method == X265GPU_DIA)
        hipLaunchKernelGGL((search_flow_kernel<P, X265GPU_DIA, STAGED>), grid, dim3(256), 0, s, dJobs, n, g, level0, chunkTab,
                           costTab, maxMvd, sync);
    else if (method == X265GPU_UMH)
        hipLaunchKernelGGL((search_flow_kernel<P, X265GPU_UMH, STAGED>), grid, dim3(256), 0, s, dJobs, n, g, level0, chunkTab,
                           costTab, maxMvd, sync);
    else
        hipLaunchKernelGGL((search_flow_kernel<P, X265GPU_HEX, STAGED>), grid, dim3(256), 0, s, dJobs, n, g, level0, chunkTab,
                           costTab, maxMvd, sync);
}

// Launches of up to X265GPU_FLOW_STAGED_JOBS searches (default 64, 0: never) take search_block's STAGED form: the
// neighbour vectors as they come, and the search run ahead on assumed ones.  Their time is the length of the dependent
// chain, which that form shortens at the price of more instructions and registers per block; larger launches fill
// the chip and are paid per instruction.  The default is the crossover on benchmarks/launch_size_sweep.py.  Read per
// launch: tests switch it between calls.
static int flow_staged_jobs()
{
    const char* e = getenv("X265GPU_FLOW_STAGED_JOBS");
    return e ? atoi(e) : 64;
}

hipError_t launch_search_flow(hipStream_t s, const Geometry& g, int bpp, const SearchJob* dJobs, int n, int level0, int method,
                              const uint32_t* dChunkTab, int numChunks, const uint16_t* costTab, int maxMvd, int* dSync,
                              int gridPad, int* launches)
{
    if (n <= 0)
        return hipSuccess;
    const int blocks = level0 ? g.W4 * g.H4 : g.cuCount;
    hipLaunchKernelGGL(search_flow_fill_kernel, dim3((blocks + 255) / 256, n), dim3(256), 0, s, dJobs, blocks);
    const int gx = gridPad > 1 ? (n + gridPad - 1) / gridPad * gridPad : n;
    const dim3 grid(gx, numChunks);
    const bool staged = n <= flow_staged_jobs();
    if (bpp == 1 && staged)
        launch_flow_method<uint8_t, true>(method, s, grid, dJobs, n, g, level0, dChunkTab, costTab, maxMvd, dSync);
    else if (bpp == 1)
        launch_flow_method<uint8_t, false>(method, s, grid, dJobs, n, g, level0, dChunkTab, costTab, maxMvd, dSync);
    else if (staged)
        launch_flow_method<uint16_t, true>(method, s, grid, dJobs, n, g, level0, dChunkTab, costTab, maxMvd, dSync);
    else
        launch_flow_method<uint16_t, false>(method, s, grid, dJobs, n, g, level0, dChunkTab, costTab, maxMvd, dSync);
    if (launches)
        (*launches) += 2;
    return hipGetLastError();
}

template <typename P, int METHOD>
static void launch_diag(hipStream_t s, dim3 grid, const SearchJob* dJobs, const Geometry& g, int level0, int t,
                        const uint16_t* costTab, int maxMvd)
{
    hipLaunchKernelGGL((search_diag_kernel<P, METHOD>), grid, dim3(256), 0, s, dJobs, g, level0, t, costTab, maxMvd);
}

template <typename P>
static void launch_diag_method(int method, hipStream_t s, dim3 grid, const SearchJob* dJobs, const Geometry& g,
                               int level0, int t, const uint16_t* costTab, int maxMvd)
{
    if (method == X265GPU_DIA)
        launch_diag<P, X265GPU_DIA>(s, grid, dJobs, g, level0, t, costTab, maxMvd);
    else if (method == X265GPU_UMH)
        launch_diag<P, X265GPU_UMH>(s, grid, dJobs, g, level0, t, costTab, maxMvd);
    else
        launch_diag<P, X265GPU_HEX>(s, grid, dJobs, g, level0, t, costTab, maxMvd);
}

template <typename P>
static void launch_rows_method(int method, hipStream_t s, int blocks, const SearchJob* dJobs, int n, const Geometry& g,
                               int level0, const uint16_t* costTab, int maxMvd, int* sync)
{
    if (method == X265GPU_DIA)
        hipLaunchKernelGGL((search_rows_kernel<P, X265GPU_DIA>), dim3(blocks), dim3(256), 0, s, dJobs, n, g, level0, costTab,
                           maxMvd, sync);
    else if (method == X265GPU_UMH)
        hipLaunchKernelGGL((search_rows_kernel_wide<P, X265GPU_UMH>), dim3(blocks), dim3(256), 0, s, dJobs, n, g, level0, costTab,
                           maxMvd, sync);
    else
        hipLaunchKernelGGL((search_rows_kernel<P, X265GPU_HEX>), dim3(blocks), dim3(256), 0, s, dJobs, n, g, level0, costTab,
                           maxMvd, sync);
}

size_t search_rows_sync_bytes(const Geometry& g, int n)
{
    const int H = g.H > g.H4 ? g.H : g.H4;
    return sizeof(int) * ((size_t)SYNC_ROWS + (size_t)n * H);
}

hipError_t launch_search_rows(hipStream_t s, const Geometry& g, int bpp, const SearchJob* dJobs, int n, int level0,
                              int method, const uint16_t* costTab, int maxMvd, int* dSync, int* launches)
{
    if (n <= 0)
        return hipSuccess;
    const int H = level0 ? g.H4 : g.H;
    // the error word survives from call to call; head and row counts start at zero
    hipError_t e = hipMemsetAsync(dSync + SYNC_HEAD, 0, sizeof(int) * 64, s);
    if (e == hipSuccess)
        e = hipMemsetAsync(dSync + SYNC_ROWS, 0, sizeof(int) * (size_t)n * H, s);
    if (e != hipSuccess)
        return e;
    const int items = n * H;
    int blocks = (items + 3) / 4;
    if (blocks > 2048)
        blocks = 2048;          // 8 workgroups on each of 256 compute units; more would only queue
    if (bpp == 1)
        launch_rows_method<uint8_t>(method, s, blocks, dJobs, n, g, level0, costTab, maxMvd, dSync);
    else
        launch_rows_method<uint16_t>(method, s, blocks, dJobs, n, g, level0, costTab, maxMvd, dSync);
    if (launches)
        (*launches)++;
    return hipGetLastError();
}

hipError_t launch_search(hipStream_t s, const SearchStreams& aux, const Geometry& g, int bpp, const SearchJob* dJobs, int n,
                         int level0, int method, const uint16_t* costTab, int maxMvd, int slicesMax, int sliceHeightMax,
                         int* launches)
{
    if (n <= 0)
        return hipSuccess;
    // job groups: group 0 on the main stream, the others on side streams.  Jobs of one
    // call never read each other's vectors, so the groups only meet at fork and join.
    const int groups = (n >= 2 * (aux.count + 1)) ? aux.count + 1 : 1;
    if (groups > 1)
    {
        hipError_t e = hipEventRecord(aux.fork, s);
        for (int k = 0; k + 1 < groups && e == hipSuccess; k++)
            e = hipStreamWaitEvent(aux.side[k], aux.fork, 0);
        if (e != hipSuccess)
            return e;
    }
    // the diagonals of the tallest slice cover every job's; shorter slices' waves leave at once
    const int W = level0 ? g.W4 : g.W, Hf = level0 ? g.H4 : g.H;
    const int H = (sliceHeightMax > 0 && sliceHeightMax < Hf) ? sliceHeightMax : Hf;
    const int nz = slicesMax > 1 ? slicesMax : 1;
    const int tmax = (W - 1) + 2 * (H - 1);
    for (int t = 0; t <= tmax; t++)
    {
        const int ylo = (t - (W - 1) + 1) >> 1 > 0 ? (t - (W - 1) + 1) >> 1 : 0;
        const int yhi = (t >> 1) < H - 1 ? (t >> 1) : H - 1;
        const int cnt = yhi - ylo + 1;
        if (cnt <= 0)
            continue;
        for (int k = 0; k < groups; k++)
        {
            const int j0 = (int)((long long)n * k / groups), j1 = (int)((long long)n * (k + 1) / groups);
            hipStream_t st = k ? aux.side[k - 1] : s;
            dim3 grid((cnt + 3) / 4, j1 - j0, nz);
            if (bpp == 1)
                launch_diag_method<uint8_t>(method, st, grid, dJobs + j0, g, level0, t, costTab, maxMvd);
            else
                launch_diag_method<uint16_t>(method, st, grid, dJobs + j0, g, level0, t, costTab, maxMvd);
        }
        if (launches)
            (*launches) += groups;
    }
    hipError_t e = hipGetLastError();
    for (int k = 0; k + 1 < groups && e == hipSuccess; k++)
    {
        e = hipEventRecord(aux.join[k], aux.side[k]);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(s, aux.join[k], 0);
    }
    return e;
}

} // namespace xg
