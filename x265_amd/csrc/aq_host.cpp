// Host half of adaptive quantisation: the double-precision mapping from block
// energies to QP offsets and quantiser scale factors.
//
// Follows LookaheadTLD::calcAdaptiveQuantFrame
// (source/encoder/slicetype.cpp:539-643) operation by operation, including where
// the reference mixes float constants into double expressions, because
// x265_exp2fix8 truncates and a last-bit difference can move a scale factor.
// The pixel work (block energies, luma sums, frame sums) is done on the device
// before this runs.
#include "kernels.h"
#include "../../include/x265gpu.h"

#include <atomic>
#include <limits>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

// every product and sum below rounds on its own, as the reference's scalar build does
#pragma STDC FP_CONTRACT OFF

namespace xg {

namespace {

// 256 * (2^(i/64) - 1) rounded: the 64-entry fraction table behind x265_exp2fix8
// (x265_exp2_lut, source/common/constants.cpp:552-558)
struct Exp2Table
{
    int v[64];
    Exp2Table()
    {
        for (int i = 0; i < 64; i++)
            v[i] = (int)lround(256.0 * (exp2(i / 64.0) - 1.0));
    }
};
const Exp2Table exp2Table;

// the luma-average steps of the HDR10 option (slicetype.cpp:613-635); block sums are of `samples` pixels
double hdr10_shift(uint32_t lumaSum, uint32_t samples)
{
    const uint32_t avg = lumaSum / samples;
    static const struct { uint32_t below; int add; } steps[] = {
        { 301, 3 }, { 367, 2 }, { 434, 1 }, { 501, 0 }, { 567, -1 }, { 634, -2 }, { 701, -3 }, { 767, -4 }, { 834, -5 } };
    for (const auto& s : steps)
        if (avg < s.below)
            return s.add;
    return -6;
}

// A block's activity (energy * depthScale + 1)^0.1 exactly as the host's libm returns it: pow() computes it the first
// time an energy is seen at a bit depth and the bits are kept.  Block energies are small integers (sixteenth- and
// sixty-fourth-variances of 16x16 blocks) that recur within a picture and from picture to picture of a sequence:
// after a few pictures nine lookups in ten are answered from the table.  One table per bit depth, shared by every
// thread and context of the process: an entry is written once with the one value pow() can return for it (an
// aligned 8-byte store), so a race between two threads only computes it twice.  Pages nobody touches cost nothing.
constexpr uint32_t ACTIVITY_MEMO = 1u << 21;            // energies beyond (rare: the busiest 16x16 blocks) go to pow()
std::atomic<std::atomic<uint64_t>*> activityMemo[3];     // depth 8, 10, 12
std::atomic<uint64_t> activityLookups{0}, activityHits{0};

inline double block_activity(uint32_t e, int depth, double depthScale, uint64_t& lookups, uint64_t& hits)
{
    const int d = depth == 8 ? 0 : depth == 10 ? 1 : depth == 12 ? 2 : -1;
    lookups++;
    if (d < 0 || e >= ACTIVITY_MEMO)
        return pow(e * depthScale + 1, 0.1);
    std::atomic<uint64_t>* tab = activityMemo[d].load(std::memory_order_acquire);
    if (!tab)
    {
        std::atomic<uint64_t>* fresh = (std::atomic<uint64_t>*)calloc(ACTIVITY_MEMO, sizeof(std::atomic<uint64_t>));
        if (!fresh)
            return pow(e * depthScale + 1, 0.1);
        if (activityMemo[d].compare_exchange_strong(tab, fresh, std::memory_order_acq_rel))
            tab = fresh;
        else
            free(fresh);
    }
    const uint64_t bits = tab[e].load(std::memory_order_relaxed);
    double a;
    if (bits)                                   // no activity is +0.0: the result is at least 1
    {
        hits++;
        memcpy(&a, &bits, sizeof(a));
        return a;
    }
    a = pow(e * depthScale + 1, 0.1);
    uint64_t out;
    memcpy(&out, &a, sizeof(out));
    tab[e].store(out, std::memory_order_relaxed);
    return a;
}

} // namespace

void aq_activity_memo_stats(uint64_t* lookups, uint64_t* hits)
{
    if (lookups)
        *lookups = activityLookups.load();
    if (hits)
        *hits = activityHits.load();
}

// x265_exp2fix8 (source/common/common.cpp:96-103): 2^(-x/6) in 8.8 fixed point
int aq_exp2fix8(double x)
{
    const int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0)
        return 0;
    if (i > 1023)
        return 0xffff;
    return (exp2Table.v[i & 63] + 256) << (i >> 6) >> 8;
}

// One pixel's gradient angle in whole degrees as computeEdge stores it (source/encoder/slicetype.cpp:145-153):
// float atan2 (the overload C++ picks for two floats), times 180 in float, divided by the reference's PI constant in
// double, back to float, negative angles folded by adding 180, truncated.
int edge_theta(int gradientV, int gradientH)
{
    const float radians = atan2f((float)gradientV, (float)gradientH);
    float theta = (float)((radians * 180) / 3.14159265);
    if (theta < 0)
        theta = 180 + theta;
    return (int)theta;
}

void aq_frame_offsets(const AqFrameParams& p, int blocks, const uint32_t* energy, const uint32_t* lumaSum,
                      const float* quantOffsets, double* qpAqOffset, double* qpCuTreeOffset, int32_t* invQscale,
                      const uint32_t* edgeDensity, const uint32_t* avgAngle)
{
    // slicetype.cpp:461-474
    const float modeOneConst = p.blockSize == 8 ? 11.427f : 14.427f, modeTwoConst = p.blockSize == 8 ? 8.f : 11.f;
    double strength, avg = 0, bias = 0;
    const bool autoVariance = p.aqMode == 2 || p.aqMode == 3 || p.aqMode == 4;
    std::vector<unsigned char> inclined;
    if (p.aqMode == 4)
        inclined.assign((size_t)blocks, 0);
    if (autoVariance)
    {
        // first pass (:541-578): activity = (energy + 1)^0.1, kept in qpCuTreeOffset, and its first two moments
        const double depthScale = 1.f / (1 << (2 * (p.depth - 8)));
        double sum = 0, sumSq = 0;
        uint64_t lookups = 0, hits = 0;
        for (int i = 0; i < blocks; i++)
        {
            // edge mode (:548-565): a block with edges is rated by its edge density, and marked when its mean gradient
            // angle lies within 15 degrees of a diagonal
            uint32_t e = energy[i];
            if (p.aqMode == 4 && edgeDensity[i])
            {
                e = edgeDensity[i];
                inclined[i] = (avgAngle[i] >= 30 && avgAngle[i] <= 60) || (avgAngle[i] >= 120 && avgAngle[i] <= 150);
            }
            const double a = block_activity(e, p.depth, depthScale, lookups, hits);
            qpCuTreeOffset[i] = a;
            sum += a;
            sumSq += a * a;
        }
        activityLookups.fetch_add(lookups, std::memory_order_relaxed);
        activityHits.fetch_add(hits, std::memory_order_relaxed);
        sum /= p.blockCount;
        sumSq /= p.blockCount;
        strength = p.strength * sum;
        avg = sum - 0.5f * (sumSq - modeTwoConst) / sum;
        bias = p.strength;
    }
    else
        strength = p.strength * 1.0397f;

    const float modeOneOffset = modeOneConst + 2 * (p.depth - 8);
    for (int i = 0; i < blocks; i++)
    {
        double q;
        if (p.aqMode == 3)
        {
            const double a = qpCuTreeOffset[i];
            q = strength * (a - avg) + bias * (1.f - modeTwoConst / (a * a));
        }
        else if (p.aqMode == 2)
            q = strength * (qpCuTreeOffset[i] - avg);
        else if (p.aqMode == 4)
        {
            const double a = qpCuTreeOffset[i];                 // :598-606, AQ_EDGE_BIAS 0.5
            q = inclined[i] && (a - avg > 0) ? (strength + 0.5) * (a - avg) : strength * (a - avg);
        }
        else
        {
            const uint32_t e = energy[i] > 1 ? energy[i] : 1;
            q = strength * (log2((double)e) - modeOneOffset);
        }
        if (p.hdr10)
            q += hdr10_shift(lumaSum[i], (uint32_t)(p.blockSize * p.blockSize));
        if (quantOffsets)
            q += quantOffsets[i];
        qpAqOffset[i] = q;
        qpCuTreeOffset[i] = q;
        invQscale[i] = aq_exp2fix8(q);
    }
}

// Lookahead::calcMotionAdaptiveQuantFrame from the displacements on (source/encoder/slicetype.cpp:3636-3662): the
// power, the two running sums in block order, the deviation, and the update of the blocks more than one deviation
// above the mean.  pow() of equal displacements (a panning picture has few distinct vectors) is looked up in a small
// table of the values libm returned earlier in this call.
int aq_motion_offsets(int blocks, const double* displacement, double* qpAqMotionOffset, double* qpAqOffset,
                      double* qpCuTreeOffset, int32_t* invQscale)
{
    constexpr int SLOTS = 1024;
    uint64_t keys[SLOTS];
    double vals[SLOTS];
    for (int i = 0; i < SLOTS; i++)
        keys[i] = ~0ull;                 // no displacement has this bit pattern (a NaN)
    double avg = 0, avgPow2 = 0;
    for (int i = 0; i < blocks; i++)
    {
        uint64_t bits;
        memcpy(&bits, &displacement[i], sizeof(bits));
        const int at = (int)((bits * 0x9E3779B97F4A7C15ull) >> 54);
        if (keys[at] != bits)
        {
            keys[at] = bits;
            vals[at] = pow(displacement[i], 0.1);
        }
        const double q = vals[at];
        qpAqMotionOffset[i] = q;
        avg += q;
        avgPow2 += q * q;
    }
    avg /= blocks;
    avgPow2 /= blocks;
    const double sd = sqrt((avgPow2 - (avg * avg)));
    int changed = 0;
    if (sd > 0)
        for (int i = 0; i < blocks; i++)
        {
            const double q = (qpAqMotionOffset[i] - avg) / sd;
            if (q > 1)
            {
                qpAqOffset[i] += q;
                qpCuTreeOffset[i] += q;
                invQscale[i] += aq_exp2fix8(q);
                changed++;
            }
        }
    return changed;
}

// LookaheadTLD::xPreanalyze for one layer (source/encoder/slicetype.cpp:358-428) from the device's quadrant
// sums, then xPreanalyzeQp for it (:322-335).  Parts are visited in raster order and written through a running
// pointer, as there; the mean divides by the layer's allocated part count.
double hevc_aq_layer(int width, int height, int partW, int partH, int numPartInWidth, int numPartInHeight,
                     double qpAdaptationRange, const unsigned long long* quadrants, double* dActivity, double* dQpOffset,
                     double* dCuTreeOffset)
{
    double dSumAct = 0.0;
    int k = 0;
    for (int y = 0; y < height; y += partH)
    {
        const int ch = partH < height - y ? partH : height - y;
        for (int x = 0; x < width; x += partW, k++)
        {
            const int cw = partW < width - x ? partW : width - x;
            const uint32_t numPix = (uint32_t)(cw >> 1) * (uint32_t)(ch >> 1);
            double dMinVar = 1.7e+308;                      // MAX_DOUBLE (source/common/common.h:157)
            if (numPix != 0)
            {
                for (int i = 0; i < 4; i++)
                {
                    const double dAverage = double(quadrants[(size_t)k * 8 + i]) / numPix;
                    const double dVariance = double(quadrants[(size_t)k * 8 + 4 + i]) / numPix - dAverage * dAverage;
                    dMinVar = dMinVar < dVariance ? dMinVar : dVariance;
                }
            }
            else
                dMinVar = 0.0;
            const double dAct = 1.0 + dMinVar;
            dActivity[k] = dAct;
            dSumAct += dAct;
        }
    }
    const double dAvgAct = dSumAct / (numPartInWidth * numPartInHeight);
    const double dMaxQScale = pow(2.0, qpAdaptationRange / 6.0);
    for (int i = 0; i < k; i++)
    {
        const double dNormAct = (dMaxQScale * dActivity[i] + dAvgAct) / (dActivity[i] + dMaxQScale * dAvgAct);
        const double dQp = (log2(dNormAct) / log2(2.0)) * 6.0;
        dQpOffset[i] = dQp;
        dCuTreeOffset[i] = dQp;
    }
    return dAvgAct;
}

// wp_ssd with the mean removed over the picture rounded to the nearest multiple
// of 16 each way, 4:2:0 (slicetype.cpp:672-685)
void aq_finish_wp(int width, int height, unsigned long long sum[3], unsigned long long ssd[3])
{
    const int w = ((width + 8) >> 4) << 4, h = ((height + 8) >> 4) << 4;
    for (int i = 0; i < 3; i++)
    {
        const int pw = i ? w >> 1 : w, ph = i ? h >> 1 : h;
        const unsigned long long area = (unsigned long long)(pw * ph);
        ssd[i] = ssd[i] - (sum[i] * sum[i] + (unsigned long long)((pw * ph) / 2)) / area;
    }
}

// log2 of a non-negative integer, exactly as the host's libm returns it: the value is computed by log2() the first
// time it is asked for and remembered per host thread.  The cuTree ratios take two logarithms per block and frame
// of small integers (scaled intra and propagated costs) that recur from frame to frame; the memo turns most of them
// into a load without changing one bit of the result.
static inline double log2_of_int(int x)
{
    constexpr unsigned MEMO = 1u << 18;
    static thread_local std::vector<double> memo;
    if (memo.empty())
        memo.assign(MEMO, std::numeric_limits<double>::quiet_NaN());
    if ((unsigned)x < MEMO)
    {
        double r = memo[(unsigned)x];
        if (r != r)
            memo[(unsigned)x] = r = log2((double)x);
        return r;
    }
    return log2((double)x);
}

// Lookahead::cuTreeFinish, the qgSize > 8 loop (source/encoder/slicetype.cpp:4052-4061): blocks
// whose scaled intra cost is zero keep their previous offset.
void cutree_finish_offsets(int blocks, const int32_t* intraCost, const int32_t* invQscale, const unsigned long long* totals,
                           int fpsFactor, double weightDelta, double strength, const double* qpAqOffset,
                           double* qpCuTreeOffset)
{
    for (int i = 0; i < blocks; i++)
    {
        const int intra = (intraCost[i] * invQscale[i] + 128) >> 8;
        if (!intra)
            continue;
        const int inherited = (int)(totals[i] < 65535 ? totals[i] : 65535);
        const int propagated = (inherited * fpsFactor + 128) >> 8;
        const double ratio = log2_of_int(intra + propagated) - log2_of_int(intra) + weightDelta;
        qpCuTreeOffset[i] = qpAqOffset[i] - strength * ratio;
    }
}

// Lookahead::cuTreeFinish, the qgSize 8 loop (source/encoder/slicetype.cpp:4030-4049): one ratio per lowres
// 8x8 block from a quarter of its intra and propagated costs, applied to the four full-resolution 8x8 blocks
// it covers (offset arrays of 2W x 2H entries).
void cutree_finish_offsets_qg8(int W, int H, const int32_t* intraCost, const int32_t* invQscale8x8,
                               const unsigned long long* totals, int fpsFactor, double weightDelta, double strength,
                               const double* qpAqOffset, double* qpCuTreeOffset)
{
    const int rowFull = 2 * W;
    for (int cuY = 0; cuY < H; cuY++)
        for (int cuX = 0; cuX < W; cuX++)
        {
            const int i = cuX + cuY * W;
            const int intra = (intraCost[i] / 4 * invQscale8x8[i] + 128) >> 8;
            if (!intra)
                continue;
            const int inherited = (int)(totals[i] < 65535 ? totals[i] : 65535);
            const int propagated = (inherited / 4 * fpsFactor + 128) >> 8;
            const double ratio = log2_of_int(intra + propagated) - log2_of_int(intra) + weightDelta;
            const int at = cuX * 2 + cuY * W * 4;
            qpCuTreeOffset[at] = qpAqOffset[at] - strength * ratio;
            qpCuTreeOffset[at + 1] = qpAqOffset[at + 1] - strength * ratio;
            qpCuTreeOffset[at + rowFull] = qpAqOffset[at + rowFull] - strength * ratio;
            qpCuTreeOffset[at + rowFull + 1] = qpAqOffset[at + rowFull + 1] - strength * ratio;
        }
}

// Lookahead::computeCUTreeQpOffset (source/encoder/slicetype.cpp:3876-4014), what cuTreeFinish runs with
// rc.hevcAq: for every AQ layer, dCuTreeOffset = dQpOffset - strength * (mean log2 ratio of the AQ blocks the
// part covers).  With qgSize 8 the ratios are first spread over the full-resolution 8x8 grid (ratio8x8, read-
// modify-write: blocks with a zero scaled intra cost keep what is there); otherwise they are formed per 16x16
// block inside the layer loop.
void cutree_finish_hevc(int W, int H, int widthFullRes, int heightFullRes, bool qg8, const int32_t* intraCost,
                        const int32_t* invQscale, const unsigned long long* totals, int fpsFactor, double weightDelta,
                        double strength, int numLayers, const int* partW, const int* partH, const int* numW, const int* numH,
                        const double* const* dQpOffset, double* const* dCuTreeOffset, double* ratio8x8)
{
    const int loopIncr = qg8 ? 8 : 16;
    const int maxCols = qg8 ? 2 * W : W;
    if (qg8)
    {
        for (int cuY = 0; cuY < H; cuY++)
            for (int cuX = 0; cuX < W; cuX++)
            {
                const int i = cuX + cuY * W;
                const int intra = ((intraCost[i] / 4) * invQscale[i] + 128) >> 8;
                if (!intra)
                    continue;
                const int inherited = (int)(totals[i] < 65535 ? totals[i] : 65535);
                const int propagated = ((inherited / 4) * fpsFactor + 128) >> 8;
                const double ratio = log2_of_int(intra + propagated) - log2_of_int(intra) + weightDelta;
                const int at = cuX * 2 + cuY * W * 4;
                ratio8x8[at] = ratio8x8[at + 1] = ratio8x8[at + maxCols] = ratio8x8[at + maxCols + 1] = ratio;
            }
    }
    for (int d = 0; d < numLayers; d++)
    {
        const double* qp = dQpOffset[d];
        double* out = dCuTreeOffset[d];
        for (int y = 0; y < numH[d]; y++)
            for (int x = 0; x < numW[d]; x++, qp++, out++)
            {
                const uint32_t bx0 = (uint32_t)x * partW[d], by0 = (uint32_t)y * partH[d];
                uint32_t count = 0;
                double sum = 0;
                for (uint32_t by = by0; by < by0 + partH[d] && by < (uint32_t)heightFullRes; by += loopIncr)
                    for (uint32_t bx = bx0; bx < bx0 + partW[d] && bx < (uint32_t)widthFullRes; bx += loopIncr)
                    {
                        const uint32_t idx = (by / loopIncr) * maxCols + bx / loopIncr;
                        if (qg8)
                            sum += ratio8x8[idx];
                        else
                        {
                            const int intra = (intraCost[idx] * invQscale[idx] + 128) >> 8;
                            const int inherited = (int)(totals[idx] < 65535 ? totals[idx] : 65535);
                            const int propagated = (inherited * fpsFactor + 128) >> 8;
                            sum += (log2_of_int(intra + propagated) - log2_of_int(intra) + weightDelta);
                        }
                        count++;
                    }
                *out = *qp - (strength * sum) / count;
            }
    }
}

// The arithmetic collectPictureStatistics applies to the raw counts
// (source/encoder/slicetype.cpp:1688-1831): every bin starts at 1 and is scaled by 16 afterwards, chroma
// sums are scaled by 16 for the 4x4 subsampling, and the per-segment means are rounded divisions whose
// operands are taken exactly as the reference writes them - including the terms where it puts a segment's
// height remainder next to its width.
void hist_finish(int width, int height, const uint32_t* hist, const unsigned long long* sums,
                 const unsigned long long* rowVar, int maxRows, x265gpu_hist_stats_t* out)
{
    memset(out, 0, sizeof(*out));
    unsigned long long frameSum[3] = { 0, 0, 0 };
    const uint32_t qw = (uint32_t)width / 4, qh = (uint32_t)height / 4;
    for (uint32_t sx = 0; sx < 4; sx++)
        for (uint32_t sy = 0; sy < 4; sy++)
            for (int plane = 0; plane < 3; plane++)
            {
                const int cell = (int)(sx * 4 + sy) * 3 + plane;
                for (int b = 0; b < 256; b++)
                    out->picHistogram[sx][sy][plane][b] = (hist[cell * 256 + b] + 1) << 4;
                // luma segments tile the quarter-sample picture, chroma segments the full-size one
                const uint32_t pw = plane ? (uint32_t)width : qw, ph = plane ? (uint32_t)height : qh;
                const uint32_t segW = pw / 4, segH = ph / 4;
                const uint32_t wo = sx == 3 ? pw - 4 * segW : 0, ho = sy == 3 ? ph - 4 * segH : 0;
                unsigned long long sum = sums[cell];
                uint8_t mean;
                if (plane == 0)
                {
                    mean = (uint8_t)((sum + (((segW + wo) * (segW + ho)) >> 1)) / ((segW + wo) * (segH + ho)));
                    frameSum[0] += sum << 4;
                }
                else
                {
                    sum <<= 4;
                    frameSum[plane] += sum;
                    const uint32_t area = (segW + wo) * (segH + ho);
                    const uint32_t div = plane == 1 ? area >> 2 : ((segW + ho) * (segH + ho)) >> 2;
                    mean = (uint8_t)((sum + (area >> 3)) / div);
                }
                out->averageIntensityPerSegment[sx][sy][plane] = mean;
            }
    const uint32_t area = (uint32_t)width * (uint32_t)height;
    out->averageIntensity[0] = (uint8_t)((frameSum[0] + (area >> 1)) / area);
    out->averageIntensity[1] = (uint8_t)((frameSum[1] + (area >> 3)) / (area >> 2));
    out->averageIntensity[2] = (uint8_t)((frameSum[2] + (area >> 3)) / (area >> 2));

    // computePictureStatistics (:1567-1650): row means truncated to 16 bits, then the mean of those
    uint16_t* dst[3] = { &out->picAvgVariance, &out->picAvgVarianceCb, &out->picAvgVarianceCr };
    for (int plane = 0; plane < 3; plane++)
    {
        const int w = plane ? width >> 1 : width, h = plane ? height >> 1 : height, size = plane ? 4 : 8;
        unsigned long long tot = 0;
        for (int by = 0; by * size < h; by++)
            tot += (uint16_t)(rowVar[(size_t)plane * maxRows + by] / (unsigned long long)w);
        *dst[plane] = (uint16_t)(tot / (unsigned long long)h);
    }
}

} // namespace xg
