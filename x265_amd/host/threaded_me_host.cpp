// --threaded-me CTU job generator above the x265gpu C-ABI (see threaded_me_host.h)
#include "threaded_me_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

namespace x265gpu_host {

namespace {

constexpr int BC_MAX_MV = 1 << 15;          // source/encoder/bitcost.h:92
constexpr int MVP_IDX_BITS = 1;             // source/encoder/search.h
constexpr int AREA_SEARCH_RANGE = 32;       // search.cpp:234

struct PUBlock { int w, h, part; bool amp; };

// g_puLookup (source/encoder/threadedme.h:67-92): every luma PU shape of a 64x64 CTU, smallest CU first
const PUBlock kPuLookup[MAX_NUM_PU_SIZES] = {
    { 8, 4, SIZE_2NxN, false },   { 4, 8, SIZE_Nx2N, false },   { 8, 8, SIZE_2Nx2N, false },
    { 16, 4, SIZE_2NxnU, true },  { 16, 12, SIZE_2NxnD, true }, { 4, 16, SIZE_nLx2N, true },  { 12, 16, SIZE_nRx2N, true },
    { 16, 8, SIZE_2NxN, false },  { 8, 16, SIZE_Nx2N, false },  { 16, 16, SIZE_2Nx2N, false },
    { 32, 8, SIZE_2NxnU, true },  { 32, 24, SIZE_2NxnD, true }, { 8, 32, SIZE_nLx2N, true },  { 24, 32, SIZE_nRx2N, true },
    { 32, 16, SIZE_2NxN, false }, { 16, 32, SIZE_Nx2N, false }, { 32, 32, SIZE_2Nx2N, false },
    { 64, 16, SIZE_2NxnU, true }, { 64, 48, SIZE_2NxnD, true }, { 16, 64, SIZE_nLx2N, true }, { 48, 64, SIZE_nRx2N, true },
    { 64, 32, SIZE_2NxN, false }, { 32, 64, SIZE_Nx2N, false }, { 64, 64, SIZE_2Nx2N, false },
};

inline bool operator==(MV a, MV b) { return a.x == b.x && a.y == b.y; }
inline bool operator!=(MV a, MV b) { return !(a == b); }
inline bool notZero(MV a) { return (a.x | a.y) != 0; }
inline int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }

// scaleMv + CUData::scaleMvByPOCDist (source/common/cudata.cpp:104-110, 2229-2244)
MV scaleMvByPOCDist(MV in, int curPOC, int curRefPOC, int colPOC, int colRefPOC)
{
    const int diffPocD = colPOC - colRefPOC, diffPocB = curPOC - curRefPOC;
    if (diffPocD == diffPocB)
        return in;
    const int tdb = clip3(-128, 127, diffPocB), tdd = clip3(-128, 127, diffPocD);
    const int x = (0x4000 + abs(tdd / 2)) / tdd;
    const int scale = clip3(-4096, 4095, (tdb * x + 32) >> 6);
    MV out;
    out.x = clip3(-32768, 32767, (scale * in.x + 127 + (scale * in.x < 0)) >> 8);
    out.y = clip3(-32768, 32767, (scale * in.y + 127 + (scale * in.y < 0)) >> 8);
    return out;
}

// InterNeighbourMV (source/common/cudata.h:106-125), the fields this path reads
struct Neighbour { MV mv[2]; int refIdx[2]; };

struct SliceRefs
{
    int poc;
    int numRefIdx[2];
    int refPOC[2][16];
};

// CUData::getDirectPMV / getIndirectPMV (source/common/cudata.cpp:2110-2156)
bool getDirectPMV(const SliceRefs& s, MV& pmv, const Neighbour& nb, int picList, int refIdx)
{
    const int curRefPOC = s.refPOC[picList][refIdx];
    for (int i = 0; i < 2; i++, picList = !picList)
    {
        const int partRefIdx = nb.refIdx[picList];
        if (partRefIdx >= 0 && curRefPOC == s.refPOC[picList][partRefIdx])
        {
            pmv = nb.mv[picList];
            return true;
        }
    }
    return false;
}

bool getIndirectPMV(const SliceRefs& s, MV& out, const Neighbour& nb, int picList, int refIdx)
{
    const int curRefPOC = s.refPOC[picList][refIdx];
    for (int i = 0; i < 2; i++, picList = !picList)
    {
        const int partRefIdx = nb.refIdx[picList];
        if (partRefIdx >= 0)
        {
            out = scaleMvByPOCDist(nb.mv[picList], s.poc, curRefPOC, s.poc, s.refPOC[picList][partRefIdx]);
            return true;
        }
    }
    return false;
}

// CUData::getPMV (source/common/cudata.cpp:1845-1905, 1959-1983): the two AMVP candidates and the wider list of
// motion candidates; returns the size of that list.  `temporal`: neighbours[MD_COLLOCATED] for this list, or NULL
int getPMV(const SliceRefs& s, const Neighbour* nb, int picList, int refIdx, MV amvpCand[2], MV* pmv,
           const TmeTemporal* temporal)
{
    MV direct[MD_ABOVE_LEFT + 1], indirect[MD_ABOVE_LEFT + 1];
    bool validDirect[MD_ABOVE_LEFT + 1], validIndirect[MD_ABOVE_LEFT + 1];
    for (int dir = MD_LEFT; dir <= MD_ABOVE_LEFT; dir++)
    {
        validDirect[dir] = getDirectPMV(s, direct[dir], nb[dir], picList, refIdx);
        validIndirect[dir] = getIndirectPMV(s, indirect[dir], nb[dir], picList, refIdx);
    }
    int num = 0;
    if (validDirect[MD_BELOW_LEFT])
        amvpCand[num++] = direct[MD_BELOW_LEFT];
    else if (validDirect[MD_LEFT])
        amvpCand[num++] = direct[MD_LEFT];
    else if (validIndirect[MD_BELOW_LEFT])
        amvpCand[num++] = indirect[MD_BELOW_LEFT];
    else if (validIndirect[MD_LEFT])
        amvpCand[num++] = indirect[MD_LEFT];
    const bool addedLeft = num > 0;
    if (validDirect[MD_ABOVE_RIGHT])
        amvpCand[num++] = direct[MD_ABOVE_RIGHT];
    else if (validDirect[MD_ABOVE])
        amvpCand[num++] = direct[MD_ABOVE];
    else if (validDirect[MD_ABOVE_LEFT])
        amvpCand[num++] = direct[MD_ABOVE_LEFT];
    if (!addedLeft)
    {
        if (validIndirect[MD_ABOVE_RIGHT])
            amvpCand[num++] = indirect[MD_ABOVE_RIGHT];
        else if (validIndirect[MD_ABOVE])
            amvpCand[num++] = indirect[MD_ABOVE];
        else if (validIndirect[MD_ABOVE_LEFT])
            amvpCand[num++] = indirect[MD_ABOVE_LEFT];
    }
    int numMvc = 0;
    for (int dir = MD_LEFT; dir <= MD_ABOVE_LEFT; dir++)
    {
        if (validDirect[dir] && notZero(direct[dir]))
            pmv[numMvc++] = direct[dir];
        if (validIndirect[dir] && notZero(indirect[dir]))
            pmv[numMvc++] = indirect[dir];
    }
    if (num == 2)
        num -= amvpCand[0] == amvpCand[1];
    if (temporal && temporal->valid && num < 2)
        pmv[numMvc++] = amvpCand[num++] =
            scaleMvByPOCDist(temporal->mv, s.poc, s.refPOC[picList][refIdx], temporal->colPoc, temporal->colRefPoc);
    while (num < 2)
        amvpCand[num].x = amvpCand[num].y = 0, num++;
    return numMvc;
}

// Search::getBlkBits (search.cpp:4894-4940): list-selection bits of list 0, list 1 and both, which in a B slice
// depend on the part's index and on what the part before it chose (lastMode)
void getBlkBits(int partSize, bool bPSlice, int partIdx, int lastMode, uint32_t blockBit[3])
{
    if (partSize == SIZE_2Nx2N || partSize == SIZE_NxN)
    {
        blockBit[0] = bPSlice ? 1 : 3;
        blockBit[1] = 3;
        blockBit[2] = 5;
        return;
    }
    if (bPSlice)
    {
        blockBit[0] = 3;
        blockBit[1] = blockBit[2] = 0;
        return;
    }
    static const uint32_t horizontal[2][3][3] = { { { 0, 0, 3 }, { 0, 0, 0 }, { 0, 0, 0 } },
                                                  { { 5, 7, 7 }, { 7, 5, 7 }, { 9 - 3, 9 - 3, 9 - 3 } } };
    static const uint32_t vertical[2][3][3] = { { { 0, 2, 3 }, { 0, 0, 0 }, { 0, 0, 0 } },
                                                { { 5, 7, 7 }, { 7 - 2, 7 - 2, 9 - 2 }, { 9 - 3, 9 - 3, 9 - 3 } } };
    const bool horz = partSize == SIZE_2NxN || partSize == SIZE_2NxnU || partSize == SIZE_2NxnD;
    memcpy(blockBit, (horz ? horizontal : vertical)[partIdx][lastMode], 3 * sizeof(uint32_t));
}

inline int getTUBits(int idx, int numIdx) { return idx + (idx < numIdx - 1); }   // search.h:252-255

}  // namespace

ThreadedME::ThreadedME(x265gpu_ctx* ctx, const TmeParam& p)
    : m_ctx(ctx), m_p(p), m_unsupported(NULL), m_numCuInWidth(0), m_numCuInHeight(0), m_numLevels(0), m_cuQp(NULL), m_areaQp(NULL),
      m_temporal(NULL), m_setsOnDevice(0), m_searches(0), m_deviceCalls(0), m_jobs(NULL), m_results(NULL), m_jobCap(0),
      m_pool(NULL)
{
    m_seconds[0] = m_seconds[1] = m_seconds[2] = 0;
    const int ctu = p.maxCUSize, mn = p.minCUSize;
    if ((ctu != 16 && ctu != 32 && ctu != 64) || (mn != 8 && mn != 16 && mn != 32 && mn != 64) || mn > ctu)
        m_unsupported = "maxCUSize must be 16, 32 or 64 and minCUSize 8..maxCUSize";
    else if (p.picWidth < ctu || p.picHeight < ctu || p.picWidth % mn || p.picHeight % mn)
        m_unsupported = "picture size must be a multiple of minCUSize";
    else if (p.searchRange < 1 || p.searchRange > 32767 || p.subpelRefine < 0 || p.subpelRefine > 7 || !(p.lambda > 0))
        m_unsupported = "searchRange, subpelRefine or lambda out of range";
    else if (p.searchMethod == 4 /* X265_SEA */ && (p.bEnableAMP || (mn == 8 && p.bEnableRectInter) || p.searchRange > 126))
        // the elimination test of 8x4, 4x8, 32x8 and 8x32 PUs reads encode-buffer bytes outside the PU, and the lower
        // terms of 64x48, 64x16 and 32x24 are read up to 56 columns to the right, past the planes at the picture's edge
        m_unsupported = "successive elimination needs --no-amp, rectangular PUs only from 16x16 CUs up, and merange <= 126";
    if (m_unsupported)
        return;
    m_numCuInWidth = (p.picWidth + ctu - 1) / ctu;
    m_numCuInHeight = (p.picHeight + ctu - 1) / ctu;
    // BitCost::s_bitsizes and the lambda-scaled table of BitCost::setQP (source/encoder/bitcost.cpp:31-101)
    m_bitsizes.resize(4 * BC_MAX_MV + 1);
    float* bs = m_bitsizes.data() + 2 * BC_MAX_MV;
    const float twoOverLn2 = 2.0f / logf(2.0f);
    bs[0] = 0.718f;
    for (int i = 1; i <= 2 * BC_MAX_MV; i++)
        bs[i] = bs[-i] = logf((float)(i + 1)) * twoOverLn2 + 1.718f;
    addCostSet(p.lambda);
    buildTasks();
}

// the lambda-scaled table of BitCost::setQP (source/encoder/bitcost.cpp:31-60) and RDCost::setLambda's fixed-point lambda
// (source/encoder/rdcost.h:88-92) for one more lambda
void ThreadedME::addCostSet(double lambda)
{
    m_costSets.emplace_back();
    CostSet& cs = m_costSets.back();
    cs.table.resize(4 * BC_MAX_MV + 1);
    const float* bs = m_bitsizes.data() + 2 * BC_MAX_MV;
    uint16_t* ct = cs.table.data() + 2 * BC_MAX_MV;
    for (int i = 0; i <= 2 * BC_MAX_MV; i++)
    {
        const double scaled = bs[i] * lambda + 0.5f;
        ct[i] = ct[-i] = (uint16_t)(scaled < 32767.0 ? scaled : 32767.0);
    }
    cs.lambdaFix8 = (uint64_t)floor(256.0 * lambda);
}

int ThreadedME::setCuQp(const int8_t* qp, const int8_t* areaQp, const double* lambdaByQp, int numQp)
{
    if (qp && (!lambdaByQp || numQp < 1 || numQp > 128))
        return X265GPU_ERR_ARG;
    m_cuQp = qp;
    m_areaQp = qp ? areaQp : NULL;
    if (qp)
    {
        m_lambdaByQp.assign(lambdaByQp, lambdaByQp + numQp);
        if (m_setOfQp.size() != (size_t)numQp)
        {
            // another table of lambdas: start over with the sets (set 0, TmeParam's lambda, stays)
            m_costSets.resize(1);
            m_setsOnDevice = 0;
            m_setOfQp.assign((size_t)numQp, -1);
        }
    }
    return X265GPU_OK;
}

int ThreadedME::cusPerCtu() const
{
    int n = 0;
    for (int size = m_p.maxCUSize; size >= m_p.minCUSize; size >>= 1)
        n += (m_p.maxCUSize / size) * (m_p.maxCUSize / size);
    return n;
}

int ThreadedME::cuIndex(const PuTask& t) const
{
    int at = 0;
    for (int size = m_p.maxCUSize; size > t.cuSize; size >>= 1)
        at += (m_p.maxCUSize / size) * (m_p.maxCUSize / size);
    return at + (t.cuY / t.cuSize) * (m_p.maxCUSize / t.cuSize) + t.cuX / t.cuSize;
}

int ThreadedME::costSetOfQp(int qp)
{
    if (qp < 0 || (size_t)qp >= m_setOfQp.size() || !(m_lambdaByQp[qp] > 0))
        return -1;
    if (m_setOfQp[qp] < 0)
    {
        if (m_costSets.size() >= (size_t)X265GPU_PU_MAX_COST_TABLES)
            return -1;
        addCostSet(m_lambdaByQp[qp]);
        m_setOfQp[qp] = (int)m_costSets.size() - 1;
    }
    return m_setOfQp[qp];
}

// ThreadedME::initPuStartIdx (source/encoder/threadedme.cpp:84-106), then the call order of
// Analysis::computeMVForPUs for one CTU
void ThreadedME::buildTasks()
{
    const int ctu = m_p.maxCUSize;
    for (auto& row : m_puStartIdx)
        for (int& v : row)
            v = -1;
    int startIdx = 0;
    for (const PUBlock& pu : kPuLookup)
    {
        if (pu.w > ctu || pu.h > ctu)
            continue;
        const int indexW = pu.amp ? std::max(pu.w, pu.h) : pu.w, indexH = pu.amp ? indexW : pu.h;
        const int numPUs = (ctu / indexW) * (ctu / indexH);
        m_puStartIdx[pu.w + pu.h][pu.part] = startIdx;
        startIdx += pu.amp ? 2 * numPUs : numPUs;
    }
    std::vector<int> writer(MAX_NUM_PUS_PER_CTU, -1);
    m_tasks.clear();
    visit(0, 0, ctu, writer);
    m_numLevels = 0;
    for (const PuTask& t : m_tasks)
        m_numLevels = std::max(m_numLevels, t.level + 1);
}

// Analysis::computeMVForPUs (analysis.cpp:161-246): children first, then every shape of this CU size.
// writer[entry] = the task that fills that entry of the CTU's table, -1 while nobody has.
void ThreadedME::visit(int cuX, int cuY, int cuSize, std::vector<int>& writer)
{
    const int ctu = m_p.maxCUSize;
    if (cuSize > m_p.minCUSize)
        for (int sub = 0; sub < 4; sub++)
            visit(cuX + (sub & 1) * (cuSize >> 1), cuY + (sub >> 1) * (cuSize >> 1), cuSize >> 1, writer);
    const int areaId = cuSize != ctu ? (cuX >= ctu / 2) + 2 * (cuY >= ctu / 2) + 1 : 0;
    for (const PUBlock& pu : kPuLookup)
    {
        if (pu.w > cuSize || pu.h > cuSize || (pu.w != cuSize && pu.h != cuSize))
            continue;
        if (!m_p.bEnableAMP && pu.amp)
            continue;
        if (!m_p.bEnableRectInter && pu.w != pu.h && !pu.amp)
            continue;
        const int startIdx = m_puStartIdx[pu.w + pu.h][pu.part];
        const int blockW = pu.amp ? std::max(pu.w, pu.h) : pu.w, blockH = pu.amp ? blockW : pu.h;
        const int numCols = ctu / blockW, numRows = ctu / blockH;
        PuTask t;
        memset(&t, 0, sizeof(t));
        t.cuX = cuX;
        t.cuY = cuY;
        t.cuSize = cuSize;
        t.partSize = pu.part;
        t.numPart = pu.part == SIZE_2Nx2N ? 1 : 2;
        t.areaId = areaId;
        t.puOffset = pu.amp ? numRows * numCols : pu.part == SIZE_2NxN ? numCols : pu.part == SIZE_Nx2N ? 1 : 0;
        const int puCol = cuX / blockW, puRow = cuY / blockH;
        t.finalIdx = startIdx + puRow * numCols + puCol;
        const int left = puCol > 0 ? startIdx + puRow * numCols + puCol - 1 : -1;
        const int above = puRow > 0 ? startIdx + (puRow - 1) * numCols + puCol : -1;
        const int aboveLeft = puRow > 0 && puCol > 0 ? startIdx + (puRow - 1) * numCols + puCol - 1 : -1;
        const int aboveRight = puRow > 0 && puCol < numCols - 1 ? startIdx + (puRow - 1) * numCols + puCol + 1 : -1;
        const int nb[5] = { left, above, aboveRight, -1, aboveLeft };
        t.level = 0;
        for (int dir = 0; dir < 5; dir++)
        {
            // an entry nobody has filled yet still holds REF_NOT_VALID when x265 reads it (framedata.cpp:93-102)
            const bool written = nb[dir] >= 0 && writer[nb[dir]] >= 0;
            t.nb[dir] = written ? nb[dir] : -1;
            if (written)
                t.level = std::max(t.level, m_tasks[writer[nb[dir]]].level + 1);
        }
        // CUData::getPartIndexAndSize for both parts (source/common/cudata.cpp:1313-1322)
        const int s = cuSize, q = s >> 2, hlf = s >> 1;
        switch (pu.part)
        {
        case SIZE_2Nx2N: t.pu[0] = { 0, 0, s, s }; break;
        case SIZE_2NxN:  t.pu[0] = { 0, 0, s, hlf };   t.pu[1] = { 0, hlf, s, hlf }; break;
        case SIZE_Nx2N:  t.pu[0] = { 0, 0, hlf, s };   t.pu[1] = { hlf, 0, hlf, s }; break;
        case SIZE_2NxnU: t.pu[0] = { 0, 0, s, q };     t.pu[1] = { 0, q, s, s - q }; break;
        case SIZE_2NxnD: t.pu[0] = { 0, 0, s, s - q }; t.pu[1] = { 0, s - q, s, q }; break;
        case SIZE_nLx2N: t.pu[0] = { 0, 0, q, s };     t.pu[1] = { q, 0, s - q, s }; break;
        default:         t.pu[0] = { 0, 0, s - q, s }; t.pu[1] = { s - q, 0, q, s }; break;
        }
        for (int k = 0; k < t.numPart; k++)
        {
            t.pu[k].x += cuX;
            t.pu[k].y += cuY;
            writer[t.finalIdx + k * t.puOffset] = (int)m_tasks.size();
        }
        m_tasks.push_back(t);
    }
}

// BitCost::bitcost(mv, mvp) and BitCost::mvcost (source/encoder/bitcost.h:56-69)
uint32_t ThreadedME::bitcost(MV mv, MV mvp) const
{
    const float* bs = m_bitsizes.data() + 2 * BC_MAX_MV;
    return (uint32_t)(bs[mv.x - mvp.x] + bs[mv.y - mvp.y] + 0.5f);
}

uint32_t ThreadedME::mvcost(MV mv, MV mvp, int set) const
{
    const uint16_t* ct = m_costSets[set].table.data() + 2 * BC_MAX_MV;
    return (uint16_t)(ct[mv.x - mvp.x] + ct[mv.y - mvp.y]);
}

// CUData::clipMv (source/common/cudata.cpp:2094-2107)
void ThreadedME::clipMv(int cuPelX, int cuPelY, MV& mv) const
{
    const int offset = 8;
    const int xmax = (m_p.picWidth + offset - cuPelX - 1) * 4, xmin = -((m_p.maxCUSize + offset + cuPelX - 1) * 4);
    const int ymax = (m_p.picHeight + offset - cuPelY - 1) * 4, ymin = -((m_p.maxCUSize + offset + cuPelY - 1) * 4);
    mv.x = std::min(xmax, std::max(xmin, mv.x));
    mv.y = std::min(ymax, std::max(ymin, mv.y));
}

// Search::setSearchRange (search.cpp:4969-5022) for one slice, no intra refresh, and a reference that is not
// the current picture
void ThreadedME::setSearchRange(int cuPelX, int cuPelY, MV mvp, int merange, MV& mvmin, MV& mvmax) const
{
    mvmin = { mvp.x - merange * 4, mvp.y - merange * 4 };
    mvmax = { mvp.x + merange * 4, mvp.y + merange * 4 };
    clipMv(cuPelX, cuPelY, mvmin);
    clipMv(cuPelX, cuPelY, mvmax);
    const int maxMvLen = (1 << 15) - 1;
    mvmin.x = std::max(mvmin.x, -maxMvLen);
    mvmin.y = std::max(mvmin.y, -maxMvLen);
    mvmax.x = std::min(mvmax.x, maxMvLen);
    mvmax.y = std::min(mvmax.y, maxMvLen);
    mvmin.x >>= 2;
    mvmin.y >>= 2;
    mvmax.x >>= 2;
    mvmax.y >>= 2;
    mvmin.y = std::min(mvmin.y, m_p.refLagPixels);
    mvmax.y = std::min(mvmax.y, m_p.refLagPixels);
    mvmax.y = std::max(mvmax.y, mvmin.y);
}

// Host threads for the per-PU predictor and bookkeeping loops: the PUs of a level are independent, x265 itself
// spreads this work over its ThreadedME workers.  run(n, fn) calls fn(begin, end) over [0, n) in chunks.
class ThreadedME::Pool
{
public:
    explicit Pool(int threads)
    {
        for (int i = 1; i < threads; i++)
            m_threads.emplace_back([this]() { loop(); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lock(m_mutex);
            m_stop = true;
            m_generation++;
        }
        m_wake.notify_all();
        for (std::thread& t : m_threads)
            t.join();
    }
    void run(size_t n, const std::function<void(size_t, size_t)>& fn)
    {
        const size_t grain = 2048;
        if (m_threads.empty() || n <= grain)
        {
            if (n)
                fn(0, n);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(m_mutex);
            m_fn = &fn;
            m_n = n;
            m_grain = grain;
            m_next.store(0);
            m_busy = (int)m_threads.size();
            m_generation++;
        }
        m_wake.notify_all();
        work();
        std::unique_lock<std::mutex> lock(m_mutex);
        m_done.wait(lock, [this]() { return m_busy == 0; });
        m_fn = nullptr;
    }

private:
    void work()
    {
        for (size_t b = m_next.fetch_add(m_grain); b < m_n; b = m_next.fetch_add(m_grain))
            (*m_fn)(b, std::min(m_n, b + m_grain));
    }
    void loop()
    {
        unsigned long long seen = 0;
        for (;;)
        {
            {
                std::unique_lock<std::mutex> lock(m_mutex);
                m_wake.wait(lock, [&]() { return m_generation != seen; });
                seen = m_generation;
                if (m_stop)
                    return;
            }
            work();
            {
                std::lock_guard<std::mutex> lock(m_mutex);
                m_busy--;
            }
            m_done.notify_one();
        }
    }
    std::vector<std::thread> m_threads;
    std::mutex m_mutex;
    std::condition_variable m_wake, m_done;
    unsigned long long m_generation = 0;
    bool m_stop = false;
    const std::function<void(size_t, size_t)>* m_fn = nullptr;
    size_t m_n = 0, m_grain = 1;
    std::atomic<size_t> m_next{ 0 };
    int m_busy = 0;
};

ThreadedME::~ThreadedME()
{
    delete m_pool;
    if (m_jobs)
        x265gpu_host_free(m_ctx, m_jobs);
    if (m_results)
        x265gpu_host_free(m_ctx, m_results);
}

namespace {

struct Stopwatch
{
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit Stopwatch(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~Stopwatch() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// one puIdx of one puMotionEstimation call of one CTU
struct Item
{
    int ctuAddr, ctuLocal, task, puIdx;
    int x, y, w, h;              // PU in the picture
    int cuPelX, cuPelY;
    int set;                     // its CU's lambda: cost set and MV-cost table on the device
};

// what one reference's pass leaves for the closing bookkeeping
struct RefPass
{
    MV amvp[2];
    MV mvp, mvpLowres;
    int mvpIdx, numMvc;
    MV mvc[14];
    bool bLowresMVP;
    int cmpJob, job0, job1;
    MV cmpMv[2];                 // the two AMVP candidates as selectMVP measures them
    MV mvmin[2], mvmax[2];       // windows of the search from the predictor and of the one from the lookahead's vector
};

// one reference's search as it comes back, before any bits are counted
struct RefRaw { MV outmv, meMvp; int satdCost; bool bLowresMVP; MV mvp, mvpLowres, amvp[2]; int mvpIdx; };

// MotionData (source/encoder/search.h:97-117), the fields this path reads
struct BestME { MV mv, mvp; uint32_t bits, cost, mvCost; int ref; };

// the bidirectional candidate of one PU, waiting for its SATDs
struct BiReq { size_t item; int r0, r1; bool tryZero; int job, jobZero; };

inline x265gpu_pu_job makeJob(const Item& it, MV mvp, MV mvmin, MV mvmax, int numMvc, const MV* mvc)
{
    x265gpu_pu_job j;
    memset(&j, 0, sizeof(j));
    j.flags = X265GPU_PU_COST_TABLE(it.set);
    j.x = it.x;
    j.y = it.y;
    j.w = it.w;
    j.h = it.h;
    j.mvpX = mvp.x;
    j.mvpY = mvp.y;
    j.mvminX = mvmin.x;
    j.mvminY = mvmin.y;
    j.mvmaxX = mvmax.x;
    j.mvmaxY = mvmax.y;
    j.numCandidates = numMvc;
    for (int i = 0; i < numMvc; i++)
    {
        j.mvc[2 * i] = mvc[i].x;
        j.mvc[2 * i + 1] = mvc[i].y;
    }
    return j;
}

}  // namespace

int ThreadedME::estimateRows(int slotFenc, int poc, int numRefs, const TmeRef* refs, int rowBegin, int rowEnd,
                             const int32_t* medianMv, MEData* out)
{
    return estimateRowsB(slotFenc, poc, numRefs, 0, refs, rowBegin, rowEnd, medianMv, out);
}

int ThreadedME::estimateRowsB(int slotFenc, int poc, int numRefsL0, int numRefsL1, const TmeRef* refs, int rowBegin,
                              int rowEnd, const int32_t* medianMv, MEData* out)
{
    if (m_unsupported || !m_ctx || !refs || !out || numRefsL0 < 1 || numRefsL0 > 16 || numRefsL1 < 0 || numRefsL1 > 16 ||
        rowBegin < 0 || rowEnd > m_numCuInHeight || rowBegin >= rowEnd)
        return X265GPU_ERR_ARG;
    Stopwatch whole(m_seconds[2]);
    // list 0's references, then list 1's: the order puMotionEstimation visits them in (search.cpp:264-267)
    const int numRefs = numRefsL0 + numRefsL1;
    const bool isP = numRefsL1 == 0;
    auto listOf = [&](int r) { return r < numRefsL0 ? 0 : 1; };
    auto idxOf = [&](int r) { return r < numRefsL0 ? r : r - numRefsL0; };
    SliceRefs slice;
    memset(&slice, 0, sizeof(slice));
    slice.poc = poc;
    slice.numRefIdx[0] = numRefsL0;
    slice.numRefIdx[1] = numRefsL1;
    for (int r = 0; r < numRefs; r++)
    {
        if (refs[r].poc == poc)
            return X265GPU_ERR_UNSUPPORTED;          // the current picture as its own reference (screen content)
        slice.refPOC[listOf(r)][idxOf(r)] = refs[r].poc;
    }
    const int ctu = m_p.maxCUSize;
    const int numCtus = (rowEnd - rowBegin) * m_numCuInWidth, firstCtu = rowBegin * m_numCuInWidth;
    const int lowresBlocksInRow = ((m_p.picWidth / 2) + 7) >> 3;       // Lowres::maxBlocksInRow
    // the lambdas of these rows' CUs: one cost set (MV-cost table, fixed-point lambda) each; their tables go to the
    // device together whenever a new one has appeared
    const int perCtuQp = cusPerCtu();
    if (m_cuQp)
        for (int c = 0; c < numCtus; c++)
            for (int k = 0; k < perCtuQp; k++)
                if (costSetOfQp(m_cuQp[(size_t)(firstCtu + c) * perCtuQp + k]) < 0)
                    return X265GPU_ERR_ARG;
    if (m_areaQp)
        for (int k = firstCtu * 5; k < (firstCtu + numCtus) * 5; k++)
            if (costSetOfQp(m_areaQp[k]) < 0)
                return X265GPU_ERR_ARG;
    int rc = X265GPU_OK;
    if (m_setsOnDevice != m_costSets.size())
    {
        std::vector<const uint16_t*> centres;
        for (const CostSet& cs : m_costSets)
            centres.push_back(cs.table.data() + 2 * BC_MAX_MV);
        rc = x265gpu_me_set_mvcost_tables(m_ctx, (int32_t)centres.size(), centres.data(), 2 * BC_MAX_MV);
        if (rc)
            return rc;
        m_setsOnDevice = m_costSets.size();
        m_deviceCalls++;
    }
    auto setOf = [&](int ctuAddr, int cu) { return m_cuQp ? m_setOfQp[m_cuQp[(size_t)ctuAddr * perCtuQp + cu]] : 0; };

    for (int c = 0; c < numCtus; c++)
        for (int k = 0; k < MAX_NUM_PUS_PER_CTU; k++)
        {
            MEData& e = out[(size_t)(firstCtu + c) * MAX_NUM_PUS_PER_CTU + k];
            memset(&e, 0, sizeof(e));
            e.ref[0] = e.ref[1] = REF_NOT_VALID;
        }

    // ---- the area searches of deriveMVsForCTU (analysis.cpp:258-302): a diamond search over the CTU and over
    //      its four quadrants from the zero vector, per reference; their full-pel results become predictors ----
    const int numAreas = ctu != m_p.minCUSize ? 5 : 1;
    std::vector<MV> areaBest((size_t)numCtus * 5 * numRefs, MV{ 0, 0 });
    // job and result arrays in page-locked memory, kept between calls and only ever grown
    auto ensureJobs = [&](size_t n) -> int {
        if (n <= m_jobCap)
            return X265GPU_OK;
        if (m_jobs)
            x265gpu_host_free(m_ctx, m_jobs);
        if (m_results)
            x265gpu_host_free(m_ctx, m_results);
        m_jobs = NULL;
        m_results = NULL;
        m_jobCap = 0;
        const size_t cap = n + n / 4;
        int e = x265gpu_host_alloc(m_ctx, cap * sizeof(x265gpu_pu_job), (void**)&m_jobs);
        if (e == X265GPU_OK)
            e = x265gpu_host_alloc(m_ctx, cap * sizeof(x265gpu_pu_result), (void**)&m_results);
        if (e == X265GPU_OK)
            m_jobCap = cap;
        return e;
    };
    for (int r = 0; r < numRefs; r++)
    {
        std::vector<x265gpu_pu_job> jobs;
        std::vector<x265gpu_pu_result> results;
        for (int c = 0; c < numCtus; c++)
        {
            const int addr = firstCtu + c, ctuX = (addr % m_numCuInWidth) * ctu, ctuY = (addr / m_numCuInWidth) * ctu;
            for (int a = 0; a < numAreas; a++)
            {
                Item it;
                it.set = m_areaQp ? m_setOfQp[m_areaQp[(size_t)addr * 5 + a]] : setOf(addr, a);
                it.w = it.h = a ? ctu / 2 : ctu;
                it.x = ctuX + (a ? ((a - 1) & 1) * (ctu / 2) : 0);
                it.y = ctuY + (a ? ((a - 1) >> 1) * (ctu / 2) : 0);
                MV mvmin, mvmax;
                setSearchRange(it.x, it.y, MV{ 0, 0 }, AREA_SEARCH_RANGE, mvmin, mvmax);
                jobs.push_back(makeJob(it, MV{ 0, 0 }, mvmin, mvmax, 0, NULL));
            }
        }
        results.resize(jobs.size());
        {
            Stopwatch sw(m_seconds[1]);
            rc = x265gpu_pu_motion_estimate(m_ctx, slotFenc, refs[r].slot, X265GPU_AREA_DIAMOND_SEARCH, AREA_SEARCH_RANGE, 0,
                                            (int)jobs.size(), jobs.data(), results.data());
        }
        if (rc)
            return rc;
        m_deviceCalls++;
        m_searches += (int64_t)jobs.size();
        for (int c = 0; c < numCtus; c++)
            for (int a = 0; a < numAreas; a++)
            {
                const x265gpu_pu_result& res = results[(size_t)c * numAreas + a];
                areaBest[((size_t)c * 5 + a) * numRefs + r] = MV{ res.mvX, res.mvY };
            }
        if (medianMv)
            for (int c = 0; c < numCtus; c++)
            {
                const int32_t* m = medianMv + ((size_t)(firstCtu + c) * numRefs + r) * 3;
                if (m[0])
                    for (int a = 0; a < 5; a++)
                        areaBest[((size_t)c * 5 + a) * numRefs + r] = MV{ m[1], m[2] };
            }
    }

    // ---- computeMVForPUs, one dependency level at a time over all CTUs ----
    if (!m_pool)
    {
        const char* env = getenv("X265GPU_HOST_THREADS");
        const unsigned hw = std::thread::hardware_concurrency();
        int threads = env ? atoi(env) : (int)(hw ? (hw < 16 ? hw : 16) : 4);
        m_pool = new Pool(threads < 1 ? 1 : threads);
    }
    std::vector<std::vector<int> > tasksOfLevel(m_numLevels);
    for (size_t ti = 0; ti < m_tasks.size(); ti++)
        tasksOfLevel[m_tasks[ti].level].push_back((int)ti);
    std::vector<Item> items;
    std::vector<RefPass> pass;
    std::vector<RefRaw> raw;
    std::vector<BestME> bestOf;          // per item: bestME[0], bestME[1] after that part
    std::vector<int> modeOf;             // per item: lastMode after that part
    std::vector<int> biOf;               // per item: its entry in biReqs, or -1
    std::vector<unsigned char> biWant;   // per item: 0 no bidirectional candidate, 1 candidate, 2 with the zero-vector form
    std::vector<BiReq> biReqs;
    std::vector<x265gpu_bipred_job> biJobs;
    std::vector<int32_t> biCosts, biAll;
    std::vector<x265gpu_block_job> cmpJobs;
    std::vector<int32_t> cmpCosts;
    for (int level = 0; level < m_numLevels; level++)
    {
        // the level's parts of one CTU, in call order; every CTU of the rows has the same
        std::vector<std::pair<int, int> > parts;
        for (int ti : tasksOfLevel[level])
            for (int k = 0; k < m_tasks[ti].numPart; k++)
                parts.push_back(std::make_pair(ti, k));
        const size_t perCtu = parts.size(), numItems = perCtu * numCtus;
        if (items.size() < numItems)
            items.resize(numItems);
        m_pool->run(numItems, [&](size_t begin, size_t end) {
            for (size_t i = begin; i < end; i++)
            {
                const int c = (int)(i / perCtu), ti = parts[i % perCtu].first, k = parts[i % perCtu].second;
                const int addr = firstCtu + c, ctuX = (addr % m_numCuInWidth) * ctu, ctuY = (addr / m_numCuInWidth) * ctu;
                const PuTask& t = m_tasks[ti];
                items[i] = Item{ addr, c, ti, k, ctuX + t.pu[k].x, ctuY + t.pu[k].y, t.pu[k].w, t.pu[k].h, ctuX + t.cuX, ctuY + t.cuY,
                                 setOf(addr, cuIndex(t)) };
            }
        });
        if (raw.size() < numItems * numRefs)
            raw.resize(numItems * numRefs);
        if (pass.size() < numItems)
            pass.resize(numItems);
        for (int r = 0; r < numRefs; r++)
        {
            const int list = listOf(r), refIdx = idxOf(r);
            // predictors (search.cpp:269-330)
            m_pool->run(numItems, [&](size_t begin, size_t end) {
                for (size_t i = begin; i < end; i++)
                {
                    const Item& it = items[i];
                    const PuTask& t = m_tasks[it.task];
                    RefPass& ps = pass[i];
                    ps.mvp = areaBest[((size_t)it.ctuLocal * 5 + t.areaId) * numRefs + r];
                    ps.amvp[0] = ps.amvp[1] = MV{ 0, 0 };
                    ps.mvpIdx = 0;
                    ps.cmpJob = ps.job0 = ps.job1 = -1;
                    Neighbour nb[MD_ABOVE_LEFT + 1];
                    for (int dir = MD_LEFT; dir <= MD_ABOVE_LEFT; dir++)
                    {
                        if (t.nb[dir] >= 0)
                        {
                            const MEData& nd = out[(size_t)it.ctuAddr * MAX_NUM_PUS_PER_CTU + t.nb[dir]];
                            for (int l = 0; l < 2; l++)
                            {
                                nb[dir].mv[l] = nd.mv[l];
                                nb[dir].refIdx[l] = nd.ref[l];
                            }
                        }
                        else
                        {
                            nb[dir].mv[0] = nb[dir].mv[1] = MV{ 0, 0 };
                            nb[dir].refIdx[0] = nb[dir].refIdx[1] = -1;
                        }
                    }
                    MV cand[2];
                    const TmeTemporal* temporal =
                        m_temporal ? &m_temporal[((size_t)it.ctuAddr * MAX_NUM_PUS_PER_CTU + t.finalIdx + it.puIdx * t.puOffset) * 2 + list]
                                   : NULL;
                    ps.numMvc = getPMV(slice, nb, list, refIdx, cand, ps.mvc, temporal);
                    if (ps.numMvc > 0)
                    {
                        ps.amvp[0] = cand[0];
                        ps.amvp[1] = cand[1];
                        if (cand[0] != cand[1])
                        {
                            // Search::selectMVP (search.cpp:2347-2383): SAD of the prediction at each clipped candidate
                            ps.cmpJob = 0;                    // numbered below
                            for (int k = 0; k < 2; k++)
                            {
                                ps.cmpMv[k] = cand[k];
                                clipMv(it.cuPelX, it.cuPelY, ps.cmpMv[k]);
                            }
                        }
                        else
                            ps.mvp = cand[0];
                    }
                    else if (!refs[r].isIntraSlice && refs[r].ctuMV)
                    {
                        const int pos = t.finalIdx + it.puIdx * t.puOffset;
                        const MEData& md = refs[r].ctuMV[(size_t)it.ctuAddr * MAX_NUM_PUS_PER_CTU + pos];
                        const bool bi = md.ref[0] >= 0 && md.ref[1] >= 0;
                        const bool uniL0 = md.ref[0] >= 0 && md.ref[1] == REF_NOT_VALID;
                        const bool uniL1 = md.ref[1] >= 0 && md.ref[0] == REF_NOT_VALID;
                        if (uniL0)
                            ps.mvp = md.mv[0];
                        else if (uniL1)
                            ps.mvp = md.mv[1];
                        else if (bi)
                            ps.mvp = md.mv[list];
                    }
                }
            });
            int numCmp = 0;
            for (size_t i = 0; i < numItems; i++)
                if (pass[i].cmpJob == 0)
                {
                    pass[i].cmpJob = numCmp;
                    numCmp += 2;
                }
            if (numCmp)
            {
                if (cmpJobs.size() < (size_t)numCmp)
                {
                    cmpJobs.resize(numCmp);
                    cmpCosts.resize(numCmp);
                }
                m_pool->run(numItems, [&](size_t begin, size_t end) {
                    for (size_t i = begin; i < end; i++)
                        if (pass[i].cmpJob >= 0)
                            for (int k = 0; k < 2; k++)
                                cmpJobs[pass[i].cmpJob + k] = x265gpu_block_job{ items[i].x, items[i].y, items[i].w, items[i].h,
                                                                                 pass[i].cmpMv[k].x, pass[i].cmpMv[k].y, X265GPU_OP_SAD, 0 };
                });
                Stopwatch sw(m_seconds[1]);
                rc = x265gpu_block_compare(m_ctx, slotFenc, refs[r].slot, numCmp, cmpJobs.data(), cmpCosts.data());
                if (rc)
                    return rc;
                m_deviceCalls++;
            }
            // lookahead candidate, search windows, search jobs (search.cpp:332-386)
            const int diffPoc = abs(poc - refs[r].poc);
            const int32_t* lowresMvs = diffPoc > m_p.bframes + 1 ? NULL : refs[r].lowresMvs;
            m_pool->run(numItems, [&](size_t begin, size_t end) {
                for (size_t i = begin; i < end; i++)
                {
                    const Item& it = items[i];
                    RefPass& ps = pass[i];
                    if (ps.cmpJob >= 0)
                    {
                        ps.mvpIdx = (uint32_t)cmpCosts[ps.cmpJob] <= (uint32_t)cmpCosts[ps.cmpJob + 1] ? 0 : 1;
                        ps.mvp = ps.amvp[ps.mvpIdx];
                    }
                    ps.bLowresMVP = false;
                    ps.mvpLowres = MV{ 0, 0 };
                    const int blockX = it.x + (it.w >> 1), blockY = it.y + (it.h >> 1);
                    if (blockX < m_p.picWidth && blockY < m_p.picHeight)
                    {
                        MV lmv = { 0, 0 };
                        if (lowresMvs)       // Search::getLowresMV (search.cpp:2323-2343)
                        {
                            const size_t idx = (size_t)(blockY >> 4) * lowresBlocksInRow + (blockX >> 4);
                            lmv = MV{ lowresMvs[2 * idx] * 2, lowresMvs[2 * idx + 1] * 2 };
                        }
                        if (notZero(lmv))
                        {
                            ps.mvc[ps.numMvc++] = lmv;
                            ps.bLowresMVP = true;
                        }
                        ps.mvpLowres = lmv;
                    }
                    setSearchRange(it.cuPelX, it.cuPelY, ps.mvp, m_p.searchRange, ps.mvmin[0], ps.mvmax[0]);
                    ps.job0 = 0;
                    ps.job1 = -1;
                    if (ps.bLowresMVP && notZero(ps.mvpLowres) && ps.mvpLowres != ps.mvp)
                    {
                        setSearchRange(it.cuPelX, it.cuPelY, ps.mvpLowres, m_p.searchRange, ps.mvmin[1], ps.mvmax[1]);
                        ps.job1 = 0;
                    }
                }
            });
            int numJobs = 0;
            for (size_t i = 0; i < numItems; i++)
            {
                pass[i].job0 = numJobs++;
                if (pass[i].job1 == 0)
                    pass[i].job1 = numJobs++;
            }
            rc = ensureJobs((size_t)numJobs);
            if (rc)
                return rc;
            x265gpu_pu_job* jobs = m_jobs;
            const x265gpu_pu_result* results = m_results;
            m_pool->run(numItems, [&](size_t begin, size_t end) {
                for (size_t i = begin; i < end; i++)
                {
                    const RefPass& ps = pass[i];
                    jobs[ps.job0] = makeJob(items[i], ps.mvp, ps.mvmin[0], ps.mvmax[0], ps.numMvc, ps.mvc);
                    if (ps.job1 >= 0)
                        jobs[ps.job1] = makeJob(items[i], ps.mvpLowres, ps.mvmin[1], ps.mvmax[1], ps.numMvc, ps.mvc);
                }
            });
            {
                Stopwatch sw(m_seconds[1]);
                rc = x265gpu_pu_motion_estimate(m_ctx, slotFenc, refs[r].slot, m_p.searchMethod, m_p.searchRange, m_p.subpelRefine,
                                                numJobs, m_jobs, m_results);
            }
            if (rc)
                return rc;
            m_deviceCalls++;
            m_searches += (int64_t)numJobs;
            // what each search left (search.cpp:366-386); bits are counted once every reference is back
            m_pool->run(numItems, [&](size_t begin, size_t end) {
                for (size_t i = begin; i < end; i++)
                {
                    const RefPass& ps = pass[i];
                    RefRaw& rr = raw[i * numRefs + r];
                    rr.outmv = MV{ results[ps.job0].mvX, results[ps.job0].mvY };
                    rr.satdCost = results[ps.job0].cost;
                    rr.meMvp = ps.mvp;                    // the predictor MotionEstimate holds after its last search
                    rr.bLowresMVP = ps.bLowresMVP;
                    if (ps.job1 >= 0)
                    {
                        rr.bLowresMVP = false;
                        rr.meMvp = ps.mvpLowres;
                        if (results[ps.job1].cost < rr.satdCost)
                        {
                            rr.outmv = MV{ results[ps.job1].mvX, results[ps.job1].mvY };
                            rr.satdCost = results[ps.job1].cost;
                            rr.bLowresMVP = true;
                        }
                    }
                    rr.mvp = ps.mvp;
                    rr.mvpLowres = ps.mvpLowres;
                    rr.amvp[0] = ps.amvp[0];
                    rr.amvp[1] = ps.amvp[1];
                    rr.mvpIdx = ps.mvpIdx;
                }
            });
        }

        // ---- bits, costs, the best reference of each list, the bidirectional candidate, the table entry
        //      (search.cpp:269-272, 388-556).  The first parts of all partitions, then the second parts: a second
        //      part starts from the first part's bestME (declared outside the loop over the parts) and, in a B
        //      slice, its list-selection bits depend on what the first part chose. ----
        if (bestOf.size() < numItems * 2)
            bestOf.resize(numItems * 2);
        modeOf.assign(numItems, 0);
        biOf.assign(numItems, -1);
        biWant.assign(numItems, 0);
        const int zeroRange = std::max(m_p.sourceWidth ? m_p.sourceWidth : m_p.picWidth,
                                       m_p.sourceHeight ? m_p.sourceHeight : m_p.picHeight);
        for (int phase = 0; phase < 2; phase++)
        {
            m_pool->run(numItems, [&](size_t begin, size_t end) {
                for (size_t i = begin; i < end; i++)
                {
                    const Item& it = items[i];
                    if (it.puIdx != phase)
                        continue;
                    const PuTask& t = m_tasks[it.task];
                    BestME best[2];
                    int lastMode = 0;
                    if (phase)
                    {
                        best[0] = bestOf[(i - 1) * 2];
                        best[1] = bestOf[(i - 1) * 2 + 1];
                        lastMode = modeOf[i - 1];
                    }
                    else
                    {
                        memset(best, 0, sizeof(best));
                        best[0].cost = best[1].cost = 0xFFFFFFFFu;
                    }
                    uint32_t listSelBits[3];
                    getBlkBits(t.partSize, isP, it.puIdx, lastMode, listSelBits);
                    for (int r = 0; r < numRefs; r++)
                    {
                        const int list = listOf(r), refIdx = idxOf(r);
                        const RefRaw& rr = raw[i * numRefs + r];
                        uint32_t bits = listSelBits[list] + MVP_IDX_BITS + (uint32_t)getTUBits(refIdx, slice.numRefIdx[list]);
                        bits += bitcost(rr.outmv, rr.meMvp);
                        const uint32_t mvCost = mvcost(rr.outmv, rr.meMvp, it.set);
                        uint32_t cost = ((uint32_t)rr.satdCost - mvCost) + getCost(bits, it.set);
                        if (rr.bLowresMVP)
                        {
                            // Search::updateMVP (search.cpp:4961-4967)
                            const int diffBits = (int)bitcost(rr.outmv, rr.mvp) - (int)bitcost(rr.outmv, rr.mvpLowres);
                            const uint32_t origBits = bits;
                            bits = origBits + diffBits;
                            cost = (cost - getCost(origBits, it.set)) + getCost(bits, it.set);
                        }
                        // Search::checkBestMVP (search.cpp:4947-4958)
                        int mvpIdx = rr.mvpIdx;
                        const int diffBits = (int)bitcost(rr.outmv, rr.amvp[!mvpIdx]) - (int)bitcost(rr.outmv, rr.amvp[mvpIdx]);
                        if (diffBits < 0)
                        {
                            mvpIdx = !mvpIdx;
                            const uint32_t origBits = bits;
                            bits = origBits + diffBits;
                            cost = (cost - getCost(origBits, it.set)) + getCost(bits, it.set);
                        }
                        if (cost < best[list].cost)
                        {
                            best[list].mv = rr.outmv;
                            best[list].mvp = rr.amvp[mvpIdx];
                            best[list].cost = cost;
                            best[list].bits = bits;
                            best[list].mvCost = mvCost;
                            best[list].ref = refIdx;
                        }
                    }
                    bestOf[i * 2] = best[0];
                    bestOf[i * 2 + 1] = best[1];
                    // the bidirectional candidate (search.cpp:421-422): B slice, not an 8x8 CU split in two
                    // (CUData::isBipredRestriction), not 2Nx2N, a winner in both lists
                    if (!isP && !(t.cuSize == 8 && t.partSize != SIZE_2Nx2N) && t.partSize != SIZE_2Nx2N &&
                        best[0].cost != 0xFFFFFFFFu && best[1].cost != 0xFFFFFFFFu)
                    {
                        bool tryZero = notZero(best[0].mv) || notZero(best[1].mv);
                        if (tryZero)
                        {
                            // search.cpp:455-466: both predictors inside the picture-wide window around zero
                            MV mvmin, mvmax;
                            setSearchRange(it.cuPelX, it.cuPelY, MV{ 0, 0 }, zeroRange, mvmin, mvmax);
                            mvmax.y += 2;
                            mvmin = MV{ mvmin.x * 4, mvmin.y * 4 };
                            mvmax = MV{ mvmax.x * 4, mvmax.y * 4 };
                            for (int l = 0; l < 2; l++)
                                tryZero = tryZero && best[l].mvp.x >= mvmin.x && best[l].mvp.x <= mvmax.x &&
                                          best[l].mvp.y >= mvmin.y && best[l].mvp.y <= mvmax.y;
                        }
                        biWant[i] = tryZero ? 2 : 1;
                    }
                }
            });
            biReqs.clear();
            for (size_t i = 0; i < numItems; i++)
                if (items[i].puIdx == phase && biWant[i])
                {
                    BiReq q;
                    q.item = i;
                    q.r0 = bestOf[i * 2].ref;
                    q.r1 = numRefsL0 + bestOf[i * 2 + 1].ref;
                    q.tryZero = biWant[i] == 2;
                    q.job = q.jobZero = -1;
                    biOf[i] = (int)biReqs.size();
                    biReqs.push_back(q);
                }
            // the SATDs, one device call per pair of reference pictures
            biAll.assign(biReqs.size() * 2, 0);
            for (size_t first = 0; first < biReqs.size(); first++)
            {
                if (biReqs[first].job >= 0)
                    continue;
                const int r0 = biReqs[first].r0, r1 = biReqs[first].r1;
                biJobs.clear();
                for (size_t k = first; k < biReqs.size(); k++)
                {
                    BiReq& q = biReqs[k];
                    if (q.job >= 0 || q.r0 != r0 || q.r1 != r1)
                        continue;
                    const Item& it = items[q.item];
                    const BestME* best = &bestOf[q.item * 2];
                    q.job = (int)biJobs.size();
                    biJobs.push_back(x265gpu_bipred_job{ it.x, it.y, it.w, it.h, best[0].mv.x, best[0].mv.y, best[1].mv.x, best[1].mv.y });
                    if (q.tryZero)
                    {
                        q.jobZero = (int)biJobs.size();
                        biJobs.push_back(x265gpu_bipred_job{ it.x, it.y, it.w, it.h, 0, 0, 0, 0 });
                    }
                }
                biCosts.resize(biJobs.size());
                {
                    Stopwatch sw(m_seconds[1]);
                    rc = x265gpu_bipred_satd(m_ctx, slotFenc, refs[r0].slot, refs[r1].slot, (int)biJobs.size(), biJobs.data(), biCosts.data());
                }
                if (rc)
                    return rc;
                m_deviceCalls++;
                for (size_t k = first; k < biReqs.size(); k++)
                {
                    const BiReq& q = biReqs[k];
                    if (q.r0 != r0 || q.r1 != r1)
                        continue;
                    biAll[k * 2] = biCosts[q.job];
                    if (q.jobZero >= 0)
                        biAll[k * 2 + 1] = biCosts[q.jobZero];
                }
            }
            // the table entries (search.cpp:451-556)
            m_pool->run(numItems, [&](size_t begin, size_t end) {
                for (size_t i = begin; i < end; i++)
                {
                    const Item& it = items[i];
                    if (it.puIdx != phase)
                        continue;
                    const PuTask& t = m_tasks[it.task];
                    const BestME* best = &bestOf[i * 2];
                    uint32_t listSelBits[3];
                    getBlkBits(t.partSize, isP, it.puIdx, phase ? modeOf[i - 1] : 0, listSelBits);
                    uint32_t bidirCost = 0xFFFFFFFFu;
                    int bidirBits = 0;
                    MV bidirMv[2] = { best[0].mv, best[1].mv };
                    if (biOf[i] >= 0)
                    {
                        const size_t k = (size_t)biOf[i];
                        bidirBits = (int)(best[0].bits + best[1].bits + listSelBits[2] - (listSelBits[0] + listSelBits[1]));
                        bidirCost = (uint32_t)biAll[k * 2] + getCost((uint32_t)bidirBits, it.set);
                        if (biReqs[k].tryZero)
                        {
                            const MV zero = { 0, 0 };
                            const uint32_t bits0 = best[0].bits - bitcost(best[0].mv, best[0].mvp) + bitcost(zero, best[0].mvp);
                            const uint32_t bits1 = best[1].bits - bitcost(best[1].mv, best[1].mvp) + bitcost(zero, best[1].mvp);
                            const uint32_t cost = (uint32_t)biAll[k * 2 + 1] + getCost(bits0, it.set) + getCost(bits1, it.set);
                            if (cost < bidirCost)
                            {
                                bidirMv[0] = bidirMv[1] = zero;
                                bidirCost = cost;
                                bidirBits = (int)(bits0 + bits1 + listSelBits[2] - (listSelBits[0] + listSelBits[1]));
                            }
                        }
                    }
                    MEData& o = out[(size_t)it.ctuAddr * MAX_NUM_PUS_PER_CTU + t.finalIdx + it.puIdx * t.puOffset];
                    o.ref[0] = o.ref[1] = REF_NOT_VALID;
                    if (bidirCost < best[0].cost && bidirCost < best[1].cost)
                    {
                        modeOf[i] = 2;
                        for (int l = 0; l < 2; l++)
                        {
                            o.mv[l] = bidirMv[l];
                            o.mvp[l] = best[l].mvp;
                            o.mvCost[l] = best[l].mvCost;
                            o.ref[l] = best[l].ref;
                        }
                        o.bits = bidirBits;
                        o.cost = bidirCost;
                    }
                    else
                    {
                        const int l = best[0].cost <= best[1].cost ? 0 : 1;
                        modeOf[i] = l;
                        o.mv[l] = best[l].mv;
                        o.mvp[l] = best[l].mvp;
                        o.mvCost[l] = best[l].mvCost;
                        o.cost = best[l].cost;
                        o.bits = (int32_t)best[l].bits;
                        o.ref[l] = best[l].ref;
                    }
                }
            });
        }
    }
    return X265GPU_OK;
}

}  // namespace x265gpu_host

using namespace x265gpu_host;

struct xgh_tme { ThreadedME me; xgh_tme(x265gpu_ctx* c, const TmeParam& p) : me(c, p) {} };

extern "C" {

xgh_tme* xgh_tme_create(x265gpu_ctx* ctx, const xgh_tme_param* p)
{
    if (!ctx || !p)
        return NULL;
    return new (std::nothrow) xgh_tme(ctx, *p);
}

void xgh_tme_destroy(xgh_tme* t) { delete t; }

const char* xgh_tme_unsupported(const xgh_tme* t) { return t ? t->me.unsupported() : "no generator"; }

int xgh_tme_estimate_rows(xgh_tme* t, int32_t slotFenc, int32_t poc, int32_t numRefs, const xgh_tme_ref* refs,
                          int32_t rowBegin, int32_t rowEnd, const int32_t* medianMv, xgh_me_data* out)
{
    if (!t)
        return X265GPU_ERR_ARG;
    return t->me.estimateRows(slotFenc, poc, numRefs, refs, rowBegin, rowEnd, medianMv, out);
}

int xgh_tme_estimate_rows_b(xgh_tme* t, int32_t slotFenc, int32_t poc, int32_t numRefsL0, int32_t numRefsL1,
                            const xgh_tme_ref* refs, int32_t rowBegin, int32_t rowEnd, const int32_t* medianMv,
                            xgh_me_data* out)
{
    if (!t)
        return X265GPU_ERR_ARG;
    return t->me.estimateRowsB(slotFenc, poc, numRefsL0, numRefsL1, refs, rowBegin, rowEnd, medianMv, out);
}

int xgh_tme_set_cu_qp(xgh_tme* t, const int8_t* qp, const int8_t* areaQp, const double* lambdaByQp, int32_t numQp)
{
    return t ? t->me.setCuQp(qp, areaQp, lambdaByQp, numQp) : X265GPU_ERR_ARG;
}

int xgh_tme_cus_per_ctu(const xgh_tme* t) { return t ? t->me.cusPerCtu() : 0; }

void xgh_tme_set_temporal(xgh_tme* t, const xgh_tme_temporal* candidates)
{
    if (t)
        t->me.setTemporalCandidates(candidates);
}

void xgh_tme_seconds(const xgh_tme* t, double seconds[3])
{
    const double* s = t->me.seconds();
    seconds[0] = s[2] - s[1];
    seconds[1] = s[1];
    seconds[2] = s[2];
}

void xgh_tme_counts(const xgh_tme* t, int64_t counts[6])
{
    counts[0] = (int64_t)t->me.tasks().size();
    counts[1] = t->me.numLevels();
    counts[2] = t->me.numCuInWidth();
    counts[3] = t->me.numCuInHeight();
    counts[4] = t->me.searches();
    counts[5] = t->me.deviceCalls();
}
}
