// What the successive-elimination search (X265_SEA, source/encoder/motion.cpp:1438-1591) compares per partition:
// the DC of 1, 2 or 4 sub-blocks of the encode block against block sums of the reference read from one of twelve
// planes (FrameData::m_meIntegral, source/common/framedata.h:171).  The reference works this out at run time from
// the partition's size class (motion.cpp:1462-1556) and the partition's ads primitive (source/common/pixel.cpp:
// 1122-1146); the table below is the outcome per partition, shared by the host's checks and the kernel.
//
// The lower two terms of 64x48, 64x16 and 32x24 are read at a horizontal offset: the reference multiplies its row
// delta by the stride for a fixed list of partitions only (motion.cpp:1545-1550) and these are not on it.  Four
// partitions (8x4, 4x8, 32x8, 8x32) take DC terms over encode-buffer bytes outside the PU, which hold whatever an
// earlier PU left there: they have no defined result and are not in the table.
#pragma once

#ifdef __HIPCC__
#define XG_SEA_HD __host__ __device__
#else
#define XG_SEA_HD
#endif

namespace xg {

constexpr int SEA_PLANES = 12;

struct SeaShape
{
    int terms;              // 1, 2 or 4
    int plane;              // index into the twelve block-sum planes
    int tw, th;             // size of the encode sub-blocks
    int encX[4], encY[4];   // their positions inside the PU
    int sumX[4], sumY[4];   // where each term's block sum is read, relative to the candidate's origin
};

// block size of plane k: 32x32 32x24 32x8 24x32 16x16 16x12 16x4 12x16 8x32 8x8 4x16 4x4
XG_SEA_HD inline void sea_plane_size(int k, int& w, int& h)
{
    const int code = k == 0 ? 0x2020 : k == 1 ? 0x2018 : k == 2 ? 0x2008 : k == 3 ? 0x1820 : k == 4 ? 0x1010 :
                     k == 5 ? 0x100c : k == 6 ? 0x1004 : k == 7 ? 0x0c10 : k == 8 ? 0x0820 : k == 9 ? 0x0808 :
                     k == 10 ? 0x0410 : 0x0404;
    w = code >> 8;
    h = code & 0xff;
}

XG_SEA_HD inline void sea_fill(SeaShape& s, int terms, int plane, int tw, int th, int ex1, int ey1, int ex2, int ey2,
                               int ex3, int ey3, int sx1, int sy1, int sx2, int sy2, int sx3, int sy3)
{
    s.terms = terms;
    s.plane = plane;
    s.tw = tw;
    s.th = th;
    s.encX[0] = s.encY[0] = s.sumX[0] = s.sumY[0] = 0;
    s.encX[1] = ex1; s.encY[1] = ey1; s.sumX[1] = sx1; s.sumY[1] = sy1;
    s.encX[2] = ex2; s.encY[2] = ey2; s.sumX[2] = sx2; s.sumY[2] = sy2;
    s.encX[3] = ex3; s.encY[3] = ey3; s.sumX[3] = sx3; s.sumY[3] = sy3;
}

// false: the partition has no defined elimination search
XG_SEA_HD inline bool sea_shape(int w, int h, SeaShape& s)
{
    switch ((w << 8) | h)
    {
    // squares: four quarters against the quarter-size plane; 8x8 as one term
    case 0x4040: sea_fill(s, 4, 0, 32, 32, 32, 0, 0, 32, 32, 32, 32, 0, 0, 32, 32, 32); return true;
    case 0x2020: sea_fill(s, 4, 4, 16, 16, 16, 0, 0, 16, 16, 16, 16, 0, 0, 16, 16, 16); return true;
    case 0x1010: sea_fill(s, 4, 9, 8, 8, 8, 0, 0, 8, 8, 8, 8, 0, 0, 8, 8, 8); return true;
    case 0x0808: sea_fill(s, 1, 9, 8, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); return true;
    // Nx2N: upper and lower half (each a square of the PU's width) against the plane chosen by half the width
    case 0x2040: sea_fill(s, 2, 4, 32, 32, 0, 32, 0, 0, 0, 0, 0, 32, 0, 0, 0, 0); return true;
    case 0x1020: sea_fill(s, 2, 9, 16, 16, 0, 16, 0, 0, 0, 0, 0, 16, 0, 0, 0, 0); return true;
    case 0x0810: sea_fill(s, 2, 9, 8, 8, 0, 8, 0, 0, 0, 0, 0, 8, 0, 0, 0, 0); return true;
    // 2NxN: left and right half
    case 0x4020: sea_fill(s, 2, 0, 32, 32, 32, 0, 0, 0, 0, 0, 32, 0, 0, 0, 0, 0); return true;
    case 0x2010: sea_fill(s, 2, 4, 16, 16, 16, 0, 0, 0, 0, 0, 16, 0, 0, 0, 0, 0); return true;
    case 0x1008: sea_fill(s, 2, 9, 8, 8, 8, 0, 0, 0, 0, 0, 8, 0, 0, 0, 0, 0); return true;
    // asymmetric, split in four
    case 0x4030: sea_fill(s, 4, 1, 32, 24, 32, 0, 0, 24, 32, 24, 32, 0, 24, 0, 56, 0); return true;
    case 0x4010: sea_fill(s, 4, 2, 32, 8, 32, 0, 0, 8, 32, 8, 32, 0, 8, 0, 40, 0); return true;
    case 0x3040: sea_fill(s, 4, 3, 24, 32, 24, 0, 0, 32, 24, 32, 24, 0, 0, 32, 24, 32); return true;
    case 0x1040: sea_fill(s, 4, 8, 8, 32, 8, 0, 0, 32, 8, 32, 8, 0, 0, 32, 8, 32); return true;
    case 0x2018: sea_fill(s, 4, 5, 16, 12, 16, 0, 0, 12, 16, 12, 16, 0, 12, 0, 28, 0); return true;
    case 0x1820: sea_fill(s, 4, 7, 12, 16, 12, 0, 0, 16, 12, 16, 12, 0, 0, 16, 12, 16); return true;
    // asymmetric, too small to split: the whole PU's DC against a small plane
    case 0x100c: sea_fill(s, 1, 9, 16, 12, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); return true;
    case 0x0c10: sea_fill(s, 1, 11, 12, 16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); return true;
    case 0x1004: sea_fill(s, 1, 9, 16, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); return true;
    case 0x0410: sea_fill(s, 1, 11, 4, 16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); return true;
    default: return false;
    }
}

} // namespace xg
