// C-ABI layer: context, device-side Lowres mirrors ("slots"), request batching.
// See include/x265gpu.h for the contract of each entry point.
#include "../../include/x265gpu.h"
#include "kernels.h"
#include "sea_shape.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

using namespace xg;

namespace {

struct Slot
{
    char* base = nullptr;          // one allocation per slot
    char* planes = nullptr;        // 4 half-res planes
    char* planes4 = nullptr;       // 4 quarter-res planes (HME)
    int32_t* intraCost = nullptr;
    uint8_t* intraMode = nullptr;
    int32_t* invQscale = nullptr;
    bool hasAq = false;
    bool intraFresh = false;       // an intra estimate of the picture now in the slot has been launched
    unsigned long long fullSerial = 0;   // uploads of the full-resolution picture so far
    int32_t* aqStage = nullptr;    // page-locked staging copy of the caller's invQscaleFactor, allocated on first use
    hipEvent_t aqStaged = nullptr; // the last upload from aqStage
    unsigned long long* propagate = nullptr; // cuTree: unsaturated propagateCost totals, [n]
    int32_t* mvs = nullptr;        // [2][D][n][2]
    int32_t* mvCosts = nullptr;    // [2][D][n]
    int32_t* mvs4 = nullptr;       // [2][D][n4][2]
    int32_t* mvCosts4 = nullptr;   // [2][D][n4]
    uint16_t* lowresCosts = nullptr; // [D*D][n]
    int32_t* rowSatds = nullptr;     // [D*D][H]
    long long* rowPartials = nullptr;// [D*D][H][3]
    long long* sums = nullptr;       // [D*D][3]   ([0] holds the intra totals in its first two entries)
    std::vector<char*> weighted;     // [D] weighted plane sets, allocated on first use
    char* full = nullptr;            // padded full-resolution luma (config.bKeepFullRes)
    int poc = 0;
};

// Scratch of x265gpu_weight_cost_static: a fixed array in the library's device image instead of an
// allocation.  Each GPU holds its own copy at its own address; the name below is only the host-side
// handle of the symbol, so host code resolves the address once per device (resolve_static_scratch) and
// uses that pointer everywhere.  Layout: [0, 256) the cost accumulator, [256, ...) weighted plane 0.
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)8 << 20;
constexpr size_t STATIC_SCRATCH_PLANE = 256;
constexpr int STATIC_SCRATCH_MAX_DEVICES = 64;
void* g_scratchPtr[STATIC_SCRATCH_MAX_DEVICES];
// One per device: held while its address is looked up and for the length of every call that uses the array, so
// that contexts on one GPU (which all share it) take turns.
std::mutex g_scratchMutex[STATIC_SCRATCH_MAX_DEVICES];

} // namespace

static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

struct x265gpu_ctx
{
    x265gpu_config cfg;
    Geometry g;
    int bpp = 1;
    int D = 0;
    int numCoopSlices = 1;            // Lookahead::m_numCoopSlices / m_numRowsPerSlice (slicetype.cpp:1122-1133)
    int numRowsPerSlice = 0;
    std::recursive_mutex mu;          // held for the length of every entry point: calls on one context from several threads take turns
    hipStream_t stream = nullptr;
    bool ownStream = false;
    std::vector<Slot> slots;
    uint16_t* costBuf = nullptr;
    uint16_t* costCentre = nullptr;
    int maxMvd = 0;
    // staging for job descriptors and host-supplied source planes
    char* jobBuf = nullptr;
    size_t jobBufSize = 0;
    char* jobHost = nullptr;          // pinned mirror of jobBuf
    hipEvent_t jobCopied = nullptr;   // last upload out of jobHost has finished
    char* srcStage = nullptr;
    size_t srcStageSize = 0;
    unsigned int* scalar = nullptr;   // small device scratch for weight_cost
    char* staticScratch = nullptr;    // this device's copy of the library's static scratch array
    long long* batchSums = nullptr;   // [n][3] frame totals of the latest batch (device)
    long long* batchSumsHost = nullptr;
    size_t batchSumsCap = 0;
    std::string err;
    // profiling
    bool prof = false;
    double profMs[X265GPU_PHASE_COUNT] = { 0, 0, 0, 0 };
    int64_t profLaunches[X265GPU_PHASE_COUNT] = { 0, 0, 0, 0 };
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    SearchStreams searchAux;
    // full-resolution pictures (config.bKeepFullRes): PicYuv geometry, source/common/picyuv.cpp:75-103
    int fullW = 0, fullH = 0;         // interior, multiples of the CTU size
    int fullStride = 0, fullRows = 0;
    int fullMarginX = 0, fullMarginY = 0;
    uint16_t* meCostBuf = nullptr;    // BitCost table of the full-resolution searches
    uint16_t* meCostCentre = nullptr;
    int meMaxMvd = 0;
    int meCostTables = 0;             // tables resident (one per QP in use), 2 * meMaxMvd + 2 entries apart
    char* puBuf = nullptr;            // device staging: jobs, then results
    size_t puBufSize = 0;
    uint32_t* seaBuf = nullptr;       // successive elimination: 12 block-sum planes of one reference + one scratch plane
    int seaSlot = -1;                 // whose picture the planes describe, which upload of it, and which planes exist
    unsigned long long seaSerial = 0;
    unsigned seaMask = 0;
    char* mcstfBuf = nullptr;         // MCSTF motion search: vector / error arrays of every reference
    size_t mcstfBufSize = 0;
    // The intra estimate runs on a stream of its own: the motion searches that usually follow it read
    // nothing it writes, so the two overlap.  Every other entry point first makes the main stream wait.
    hipStream_t intraStream = nullptr;
    hipEvent_t intraFork = nullptr, intraDone = nullptr;
    bool intraPending = false;        // intraDone not yet awaited by the main stream
    bool intraNoFork = false;         // the next side-stream intra launch depends on nothing queued on the main stream
    char* intraJobBuf = nullptr;      // job descriptors of the intra launch in flight (device, pinned mirror)
    char* intraJobHost = nullptr;
    size_t intraJobBufSize = 0;
    // single-launch search: row-progress words on the device, and its sticky error word mirrored here
    bool searchRows = false;
    bool searchFlow = false;           // X265GPU_SEARCH_MODE=flow: one launch, vectors as ready flags
    // x265gpu_aq_frames_begin batches whose double-precision mapping is still running on host threads
    struct AqPending
    {
        std::vector<x265gpu_aq_picture> pics;
        x265gpu_aq_params params;
        char* host = nullptr;          // page-locked copy of the energy arrays (the context's aqHost / aqDev)
        char* dev = nullptr;
        size_t perPic = 0, ptrsAt = 0; // layout of those arrays: bytes per picture, where the scatter's pointers go
        std::vector<std::thread> workers;
        std::atomic<int>* next = nullptr;
    };
    std::vector<AqPending*> aqPending;
    // cuTree passes queued by x265gpu_cutree_clear / _propagate and not launched yet (cutree_flush)
    struct CuTreeOp { bool clear; unsigned long long* cells; int count; PropagateJob job; };
    std::vector<CuTreeOp> cutreeOps;
    char* cutreeBuf = nullptr;
    size_t cutreeBufBytes = 0;
    char* finishHost = nullptr;         // page-locked landing area of cuTreeFinish's three per-block arrays
    size_t finishHostBytes = 0;
    double* motionDev = nullptr;        // --aq-motion: per-block displacements (device, page-locked landing area)
    double* motionHost = nullptr;
    char* aqDev = nullptr;              // the batched AQ pass's device arrays and their page-locked mirror, kept between
    char* aqHost = nullptr;             //   calls and grown when a larger batch comes
    size_t aqBufBytes = 0;
    hipEvent_t aqInstalled = nullptr;   // the copies that install a batch's factors out of aqHost have finished
    bool aqInstallInFlight = false;     // ... recorded and not yet waited for by the host
    std::vector<int32_t> intraDeferred;    // slots whose intra estimate ran ahead of their scale factors: AQ sums to redo
    bool aqInstalledOnMain = false;        // some of those factors were uploaded on the main stream (x265gpu_lowres_set_aq)
    bool oddSize = false;             // luma-only picture of odd width or height: the 4:2:0 entry points (AQ, histogram) refuse
    int flowGridPad = 1;
    uint32_t* flowChunks[2] = { nullptr, nullptr };   // chunk tables of the half-res and the quarter-res block grid
    int flowChunkCount[2] = { 0, 0 };
    int* searchSync = nullptr;
    size_t searchSyncBytes = 0;
    int* searchErrHost = nullptr;     // pinned
    // The estimate calls whose single-launch searches nobody has checked yet (the error word is read at the calls
    // that wait for the stream).  Should a bounded wait ever run out, these are run again, in order, with a launch
    // per anti-diagonal, and the context keeps to that mode from then on.
    struct EstimateCall { std::vector<x265gpu_request> reqs; bool wantBatchOut; long long* deviceOut; };
    std::vector<EstimateCall> unverified;
    bool singleLaunchOff = false;
    int searchRecoveries = 0;
    int flowFaultAfter = -1;          // test hook (X265GPU_TEST_FLOW_FAULT): raise the error word ahead of that flow launch
    int flowLaunches = 0;
};

// x265gpu_aq_frames_begin batches still on host threads: joined, uploaded, and the intra estimates that waited for
// them launched, before anything reads scale factors or intra results
static int aq_finish_pending(x265gpu_ctx* ctx);
static int cutree_flush(x265gpu_ctx* ctx);
// work the library holds back (the AQ mapping running on host threads, queued cuTree passes) is brought up to date
// before anything that reads its results
#define FINISH_AQ(ctx) do { if (ctx) { int rcAq_ = aq_finish_pending(ctx); if (!rcAq_) rcAq_ = cutree_flush(ctx); if (rcAq_) return rcAq_; } } while (0)

namespace {

int fail(x265gpu_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
{
    if (c)
    {
        c->err = what;
        if (e != hipSuccess)
        {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIPCHK(call)                                                     \
    do                                                                   \
    {                                                                    \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess)                                            \
            return fail(ctx, X265GPU_ERR_DEVICE, #call, e_);             \
    } while (0)

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Makes the context's device current for the duration of an entry point and puts the caller's
// device back afterwards: a process may drive several contexts (one per GPU) from one thread.
// Unless told otherwise it also orders the main stream behind an intra estimate still running on its side stream.
// It also holds the context's lock, so that entry points called on one context from several host threads run one
// after the other (entry points call each other: the lock is recursive).
struct DeviceScope
{
    int prev = -1;
    std::unique_lock<std::recursive_mutex> lock;
    explicit DeviceScope(x265gpu_ctx* ctx, bool joinIntra = true);
    void release()                      // before the context, and its lock with it, is destroyed
    {
        if (lock.owns_lock())
            lock.unlock();
        lock.release();
    }
    ~DeviceScope()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// Lowres::create, source/common/lowres.cpp:84-103 and :171-177, :211-213
// `cfg` carries the picture size as the lookahead sees it in x265: Encoder::configure has padded param->sourceWidth /
// sourceHeight to the minimum CU size by then (source/encoder/encoder.cpp:4470-4474, 4689-4693), and PicYuv holds the
// source with its last column / row repeated into that padding (source/common/picyuv.cpp:572-583).  rawWidth /
// rawHeight are what the caller's planes really hold; reads beyond them repeat the edge sample.
void compute_geometry(const x265gpu_config& cfg, int rawWidth, int rawHeight, Geometry& g)
{
    memset(&g, 0, sizeof(g));
    g.srcWidth = rawWidth;
    g.srcHeight = rawHeight;
    g.marginX = cfg.maxCUSize + 32;
    g.marginY = cfg.maxCUSize + 16;
    int width = cfg.sourceWidth / 2, lines = cfg.sourceHeight / 2;
    int stride = width + 2 * g.marginX;
    if (stride & 31)
        stride += 32 - (stride & 31);
    g.stride = stride;
    g.W = (width + 7) >> 3;
    g.H = (lines + 7) >> 3;
    g.cuCount = g.W * g.H;
    g.width = g.W * 8;
    g.lines = g.H * 8;
    g.planeSize = (long long)stride * (g.lines + 2 * g.marginY);
    g.padOffset = (long long)stride * g.marginY + g.marginX;
    g.mcstfFilter = cfg.bMcstfFilter ? 1 : 0;
    // quarter-res planes exist with --hme or --mcstf (source/common/lowres.cpp:171, :388)
    if (cfg.bEnableHME || cfg.bMcstfFilter)
    {
        g.stride4 = stride / 2;
        g.width4 = g.width / 2;
        g.lines4 = g.lines / 2;
        g.marginX4 = g.marginX / 2;
        g.marginY4 = g.marginY / 2;
        g.planeSize4 = g.planeSize / 2;
        g.padOffset4 = g.padOffset / 2;
        // loop bounds: Lookahead::m_4x4Width/Height, source/encoder/slicetype.cpp:1063-1064
        g.W4 = ((cfg.sourceWidth / 4) + 7) >> 3;
        g.H4 = ((cfg.sourceHeight / 4) + 7) >> 3;
        int aw = ((g.width / 2) + 7) >> 3, ah = ((g.lines / 2) + 7) >> 3;
        g.cuCount4 = (aw > g.W4 ? aw : g.W4) * (ah > g.H4 ? ah : g.H4);
    }
}

void join_intra(x265gpu_ctx* ctx)
{
    if (ctx->intraPending)
    {
        (void)hipStreamWaitEvent(ctx->stream, ctx->intraDone, 0);
        ctx->intraPending = false;
    }
}

DeviceScope::DeviceScope(x265gpu_ctx* ctx, bool joinIntra)
{
    if (ctx)
        lock = std::unique_lock<std::recursive_mutex>(ctx->mu);
    int cur = -1;
    if (ctx && hipGetDevice(&cur) == hipSuccess && cur != ctx->cfg.deviceId && hipSetDevice(ctx->cfg.deviceId) == hipSuccess)
        prev = cur;
    if (ctx && joinIntra)
        join_intra(ctx);
}

int alloc_slot(x265gpu_ctx* ctx, Slot& s)
{
    const Geometry& g = ctx->g;
    const int D = ctx->D, n = g.cuCount, n4 = g.cuCount4;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t oPlanes = take((size_t)4 * g.planeSize * ctx->bpp);
    const size_t oPlanes4 = take((size_t)4 * g.planeSize4 * ctx->bpp);
    const size_t oIntraCost = take(sizeof(int32_t) * n);
    const size_t oIntraMode = take(n);
    const size_t oInvQ = take(sizeof(int32_t) * n);
    const size_t oPropagate = take(sizeof(unsigned long long) * n);
    const size_t oMvs = take(sizeof(int32_t) * 2 * (size_t)2 * D * n);
    const size_t oMvCosts = take(sizeof(int32_t) * (size_t)2 * D * n);
    const size_t oMvs4 = take(sizeof(int32_t) * 2 * (size_t)2 * D * n4);
    const size_t oMvCosts4 = take(sizeof(int32_t) * (size_t)2 * D * n4);
    const size_t oLowresCosts = take(sizeof(uint16_t) * (size_t)D * D * n);
    const size_t oRowSatds = take(sizeof(int32_t) * (size_t)D * D * g.H);
    const size_t oRowPartials = take(sizeof(long long) * (size_t)D * D * g.H * 3);
    const size_t oSums = take(sizeof(long long) * (size_t)D * D * 3);
    HIPCHK(hipMalloc((void**)&s.base, off));
    HIPCHK(hipMemsetAsync(s.base, 0, off, ctx->stream));
    s.planes = s.base + oPlanes;
    s.planes4 = g.planeSize4 ? s.base + oPlanes4 : nullptr;
    s.intraCost = (int32_t*)(s.base + oIntraCost);
    s.intraMode = (uint8_t*)(s.base + oIntraMode);
    s.invQscale = (int32_t*)(s.base + oInvQ);
    s.propagate = (unsigned long long*)(s.base + oPropagate);
    s.mvs = (int32_t*)(s.base + oMvs);
    s.mvCosts = (int32_t*)(s.base + oMvCosts);
    s.mvs4 = (int32_t*)(s.base + oMvs4);
    s.mvCosts4 = (int32_t*)(s.base + oMvCosts4);
    s.lowresCosts = (uint16_t*)(s.base + oLowresCosts);
    s.rowSatds = (int32_t*)(s.base + oRowSatds);
    s.rowPartials = (long long*)(s.base + oRowPartials);
    s.sums = (long long*)(s.base + oSums);
    s.weighted.assign(D, nullptr);
    if (ctx->fullStride)
        HIPCHK(hipMalloc((void**)&s.full, (size_t)ctx->fullStride * ctx->fullRows * ctx->bpp));
    return X265GPU_OK;
}

int ensure_search_sync(x265gpu_ctx* ctx, int n)
{
    const size_t bytes = align256(search_rows_sync_bytes(ctx->g, n));
    if (bytes <= ctx->searchSyncBytes)
        return X265GPU_OK;
    if (ctx->searchSync)
    {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(ctx->searchSync));
        ctx->searchSync = nullptr;
        ctx->searchSyncBytes = 0;
    }
    HIPCHK(hipMalloc((void**)&ctx->searchSync, bytes));
    HIPCHK(hipMemsetAsync(ctx->searchSync, 0, bytes, ctx->stream));
    ctx->searchSyncBytes = bytes;
    return X265GPU_OK;
}

// queue a read-back of the search's sticky error word; valid after the next stream synchronise
int fetch_search_error(x265gpu_ctx* ctx)
{
    if (ctx->searchSync)
        HIPCHK(hipMemcpyAsync(ctx->searchErrHost, ctx->searchSync + SEARCH_SYNC_ERROR, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return X265GPU_OK;
}

} // namespace

static int run_estimate(x265gpu_ctx* ctx, int32_t n, const x265gpu_request* reqs, bool wantBatchOut, long long* deviceOut);

namespace {

// Waits for the main stream and looks at the error word of the single-launch searches (flow and rows modes).  A wait
// inside them that ran out (never seen; it would mean that workgroups were not started in index order, or that another
// process held the GPU for a quarter of a second and more) left vectors behind that were searched with made-up
// predictors.  Nothing is lost: the estimate calls made since the last check are run again with one launch per
// anti-diagonal, which needs no assumption about the order in which workgroups start, and so is every later search of
// this context.  What those calls read (planes, scale factors, weighted planes) must still be in place, which holds
// for every caller that checks - waits - before it reuses a slot.
int verify_searches(x265gpu_ctx* ctx, bool* recovered = nullptr)
{
    if (recovered)
        *recovered = false;
    int rc = fetch_search_error(ctx);
    if (rc)
        return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (!ctx->searchErrHost || !ctx->searchErrHost[0])
    {
        ctx->unverified.clear();
        return X265GPU_OK;
    }
    ctx->searchErrHost[0] = 0;
    HIPCHK(hipMemsetAsync(ctx->searchSync + SEARCH_SYNC_ERROR, 0, sizeof(int), ctx->stream));
    ctx->singleLaunchOff = true;
    ctx->searchRecoveries++;
    ctx->err = "search: a wait inside the single-launch search ran out; the calls were run again with one launch per "
               "anti-diagonal, the mode this context keeps from now on";
    std::vector<x265gpu_ctx::EstimateCall> calls;
    calls.swap(ctx->unverified);
    for (const x265gpu_ctx::EstimateCall& c : calls)
    {
        rc = run_estimate(ctx, (int32_t)c.reqs.size(), c.reqs.data(), c.wantBatchOut, c.deviceOut);
        if (rc)
            return rc;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (recovered)
        *recovered = true;
    return X265GPU_OK;
}

int ensure_job_buf(x265gpu_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->jobBufSize)
        return X265GPU_OK;
    if (ctx->jobBuf)
    {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(ctx->jobBuf));
        ctx->jobBuf = nullptr;
    }
    if (ctx->jobHost)
    {
        HIPCHK(hipHostFree(ctx->jobHost));
        ctx->jobHost = nullptr;
    }
    size_t sz = align256(bytes * 2);
    HIPCHK(hipMalloc((void**)&ctx->jobBuf, sz));
    HIPCHK(hipHostMalloc((void**)&ctx->jobHost, sz, hipHostMallocDefault));
    ctx->jobBufSize = sz;
    return X265GPU_OK;
}

// Stage `bytes` of descriptors at jobBuf + offset through the pinned mirror.
// The mirror is reused by every call, so wait until the previous upload has
// drained it (kernels of earlier calls are ordered by the stream itself).
int upload_jobs(x265gpu_ctx* ctx, size_t offset, const void* src, size_t bytes, bool first)
{
    if (first)
        HIPCHK(hipEventSynchronize(ctx->jobCopied));
    memcpy(ctx->jobHost + offset, src, bytes);
    HIPCHK(hipMemcpyAsync(ctx->jobBuf + offset, ctx->jobHost + offset, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->jobCopied, ctx->stream));
    return X265GPU_OK;
}

int ensure_src_stage(x265gpu_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->srcStageSize)
        return X265GPU_OK;
    if (ctx->srcStage)
    {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(ctx->srcStage));
        ctx->srcStage = nullptr;
    }
    HIPCHK(hipMalloc((void**)&ctx->srcStage, bytes));
    ctx->srcStageSize = bytes;
    return X265GPU_OK;
}

// Job descriptors of the side-stream intra work: two halves of `intraJobBufSize` bytes each (device, and a
// page-locked mirror), the first for the estimate's launch, the second for the pass that redoes its AQ sums.
// Both record intraDone when they have been queued, so waiting for it frees both halves.
int ensure_intra_job_buf(x265gpu_ctx* ctx, size_t bytes)
{
    HIPCHK(hipEventSynchronize(ctx->intraDone));
    if (bytes <= ctx->intraJobBufSize)
        return X265GPU_OK;
    if (ctx->intraJobBuf) HIPCHK(hipFree(ctx->intraJobBuf));
    if (ctx->intraJobHost) HIPCHK(hipHostFree(ctx->intraJobHost));
    ctx->intraJobBuf = ctx->intraJobHost = nullptr;
    ctx->intraJobBufSize = 0;
    const size_t sz = align256(bytes * 2);
    HIPCHK(hipMalloc((void**)&ctx->intraJobBuf, 2 * sz));
    HIPCHK(hipHostMalloc((void**)&ctx->intraJobHost, 2 * sz, hipHostMallocDefault));
    ctx->intraJobBufSize = sz;
    return X265GPU_OK;
}

struct PhaseTimer
{
    x265gpu_ctx* c;
    int phase;
    PhaseTimer(x265gpu_ctx* ctx, int ph) : c(ctx), phase(ph)
    {
        if (c->prof)
            (void)hipEventRecord(c->ev0, c->stream);
    }
    void stop(int64_t launches)
    {
        if (!c->prof)
            return;
        (void)hipEventRecord(c->ev1, c->stream);
        (void)hipEventSynchronize(c->ev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->profMs[phase] += ms;
        c->profLaunches[phase] += launches;
    }
};

inline char* plane_origin(const x265gpu_ctx* ctx, char* planes)
{
    return planes + ctx->g.padOffset * ctx->bpp;
}
inline char* plane4_origin(const x265gpu_ctx* ctx, char* planes4)
{
    return planes4 + ctx->g.padOffset4 * ctx->bpp;
}

bool slot_ok(const x265gpu_ctx* ctx, int s) { return s >= 0 && s < (int)ctx->slots.size(); }

// Address of the static scratch array on the context's device (which must be current), looked up once
// per device and process.
int resolve_static_scratch(x265gpu_ctx* ctx)
{
    const int dev = ctx->cfg.deviceId;
    if (dev < 0 || dev >= STATIC_SCRATCH_MAX_DEVICES)
        return X265GPU_OK;              // no static scratch on such a device: weight_cost_static reports it
    std::lock_guard<std::mutex> scratchLock(g_scratchMutex[dev]);
    if (!g_scratchPtr[dev])
    {
        void* p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_scratchPtr[dev] = p;
    }
    ctx->staticScratch = (char*)g_scratchPtr[dev];
    return X265GPU_OK;
}

} // namespace

extern "C" {

int x265gpu_open(const x265gpu_config* cfg, x265gpu_ctx** out)
{
    if (!cfg || !out)
        return X265GPU_ERR_ARG;
    *out = nullptr;
    if (cfg->sourceWidth < 16 || cfg->sourceHeight < 16 || cfg->numSlots < 1 || cfg->bframes < 0 ||
        cfg->bframes > X265GPU_BFRAME_MAX ||
        (cfg->internalBitDepth != 8 && cfg->internalBitDepth != 10 && cfg->internalBitDepth != 12) ||
        (cfg->maxCUSize != 16 && cfg->maxCUSize != 32 && cfg->maxCUSize != 64))
        return X265GPU_ERR_ARG;
    auto methodOk = [](int m) { return m == X265GPU_DIA_SEARCH || m == X265GPU_HEX_SEARCH || m == X265GPU_UMH_SEARCH; };
    if (!methodOk(cfg->searchMethod))
        return X265GPU_ERR_UNSUPPORTED;
    if (cfg->bEnableHME && (!methodOk(cfg->hmeSearchMethod[0]) || !methodOk(cfg->hmeSearchMethod[1]) ||
                            cfg->hmeRange[0] < 4 || cfg->hmeRange[1] < 4))
        return X265GPU_ERR_UNSUPPORTED;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->deviceId < 0 || cfg->deviceId >= ndev)
        return X265GPU_ERR_DEVICE;
    // the caller's current device is put back when this function returns
    struct Restore
    {
        int dev = -1;
        ~Restore()
        {
            if (dev >= 0)
                (void)hipSetDevice(dev);
        }
    } restore;
    if (hipGetDevice(&restore.dev) != hipSuccess)
        restore.dev = -1;
    if (hipSetDevice(cfg->deviceId) != hipSuccess)
        return X265GPU_ERR_DEVICE;

    x265gpu_ctx* ctx = new (std::nothrow) x265gpu_ctx();
    if (!ctx)
        return X265GPU_ERR_NOMEM;
    ctx->cfg = *cfg;
    ctx->bpp = cfg->internalBitDepth > 8 ? 2 : 1;
    ctx->D = cfg->bframes + 2;
    {
        // An odd picture size (x265 takes one for 4:0:0 and 4:4:4 input only, source/common/param.cpp:1791-1794) is what
        // the caller's luma planes hold; the lookahead's geometry is that of the picture Encoder::configure padded to
        // the minimum CU size, 8 unless the caller says otherwise: a width of 16k+1 gains a column of 8x8 lowres blocks.
        x265gpu_config padded = *cfg;
        const int unit = 8;
        if (cfg->sourceWidth & 1)
            padded.sourceWidth = (cfg->sourceWidth + unit - 1) / unit * unit;
        if (cfg->sourceHeight & 1)
            padded.sourceHeight = (cfg->sourceHeight + unit - 1) / unit * unit;
        compute_geometry(padded, cfg->sourceWidth, cfg->sourceHeight, ctx->g);
        ctx->oddSize = ((cfg->sourceWidth | cfg->sourceHeight) & 1) != 0;
    }
    // cooperative slices: Lookahead::Lookahead, source/encoder/slicetype.cpp:1122-1133
    ctx->numRowsPerSlice = ctx->g.H;
    ctx->numCoopSlices = 1;
    if (cfg->lookaheadSlices > 1)
    {
        int rows = ctx->g.H / cfg->lookaheadSlices;
        rows = rows > 10 ? rows : 10;                   // at least 10 rows per slice
        rows = rows < ctx->g.H ? rows : ctx->g.H;       // but no more than the full picture
        ctx->numRowsPerSlice = rows;
        ctx->numCoopSlices = ctx->g.H / rows;
    }
    if (cfg->bKeepFullRes)
    {
        const int ctu = cfg->maxCUSize;
        ctx->fullW = (cfg->sourceWidth + ctu - 1) / ctu * ctu;
        ctx->fullH = (cfg->sourceHeight + ctu - 1) / ctu * ctu;
        ctx->fullMarginX = ctu + 32;
        ctx->fullMarginY = ctu + 16;
        ctx->fullStride = ctx->fullW + 2 * ctx->fullMarginX;
        ctx->fullRows = ctx->fullH + 2 * ctx->fullMarginY;
    }

    int rc = X265GPU_OK;
    do
    {
        if (cfg->stream)
            ctx->stream = (hipStream_t)cfg->stream;
        else
        {
            if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
            {
                rc = X265GPU_ERR_DEVICE;
                break;
            }
            ctx->ownStream = true;
        }
        if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->jobCopied, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ctx->jobCopied, ctx->stream) != hipSuccess)
        {
            rc = X265GPU_ERR_DEVICE;
            break;
        }
        {
            // side streams for the search (X265GPU_SEARCH_STREAMS = total streams, 1..4)
            const char* e = getenv("X265GPU_SEARCH_STREAMS");
            int total = e ? atoi(e) : 2;
            total = total < 1 ? 1 : (total > 4 ? 4 : total);
            SearchStreams& a = ctx->searchAux;
            bool ok = hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) == hipSuccess;
            for (int k = 0; k + 1 < total && ok; k++)
            {
                ok = hipStreamCreateWithFlags(&a.side[k], hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&a.join[k], hipEventDisableTiming) == hipSuccess;
                if (ok)
                    a.count = k + 1;
            }
            if (!ok)
            {
                rc = X265GPU_ERR_DEVICE;
                break;
            }
        }
        {
            // X265GPU_SEARCH_MODE: "flow" (default) = one launch for all anti-diagonals, blocks in diagonal order, every
            // vector its own ready flag; "diag" = one launch per anti-diagonal; "rows" = one launch with a hand-off
            // between block rows.  Requests with cooperative slices always take the diagonal launches.
            const char* e = getenv("X265GPU_SEARCH_MODE");
            ctx->searchRows = e && !strcmp(e, "rows");
            ctx->searchFlow = !e || !strcmp(e, "flow");
            const char* pad = getenv("X265GPU_FLOW_PAD");
            ctx->flowGridPad = pad ? atoi(pad) : 1;
            const char* fault = getenv("X265GPU_TEST_FLOW_FAULT");
            ctx->flowFaultAfter = fault ? atoi(fault) : -1;
            if (ctx->searchFlow)
            {
                for (int lv = 0; lv < 2 && rc == X265GPU_OK; lv++)
                {
                    const int W = lv ? ctx->g.W4 : ctx->g.W, H = lv ? ctx->g.H4 : ctx->g.H;
                    if (W <= 0 || H <= 0)
                        continue;
                    std::vector<uint32_t> tab((size_t)search_flow_chunks(W, H, nullptr));
                    search_flow_chunks(W, H, tab.data());
                    ctx->flowChunkCount[lv] = (int)tab.size();
                    if (hipMalloc((void**)&ctx->flowChunks[lv], tab.size() * sizeof(uint32_t)) != hipSuccess ||
                        hipMemcpy(ctx->flowChunks[lv], tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
                        rc = X265GPU_ERR_NOMEM;
                }
                if (rc)
                    break;
            }
            if (hipHostMalloc((void**)&ctx->searchErrHost, 64) != hipSuccess)
            {
                rc = X265GPU_ERR_NOMEM;
                break;
            }
            ctx->searchErrHost[0] = 0;
        }
        if (hipMalloc((void**)&ctx->scalar, 256) != hipSuccess)
        {
            rc = X265GPU_ERR_NOMEM;
            break;
        }
        rc = resolve_static_scratch(ctx);
        if (rc != X265GPU_OK)
            break;
        {
            // X265GPU_INTRA_STREAM=0 keeps the intra estimate on the main stream
            const char* e = getenv("X265GPU_INTRA_STREAM");
            if (hipEventCreateWithFlags(&ctx->intraFork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->intraDone, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->aqInstalled, hipEventDisableTiming) != hipSuccess ||
                hipEventRecord(ctx->intraDone, ctx->stream) != hipSuccess)
            {
                rc = X265GPU_ERR_DEVICE;
                break;
            }
            if ((!e || atoi(e) != 0) && hipStreamCreateWithFlags(&ctx->intraStream, hipStreamNonBlocking) != hipSuccess)
            {
                rc = X265GPU_ERR_DEVICE;
                break;
            }
        }
        ctx->slots.resize(cfg->numSlots);
        for (auto& s : ctx->slots)
        {
            rc = alloc_slot(ctx, s);
            if (rc != X265GPU_OK)
                break;
        }
        if (rc == X265GPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = X265GPU_ERR_DEVICE;
    } while (0);

    if (rc != X265GPU_OK)
    {
        x265gpu_close(ctx);
        return rc;
    }
    *out = ctx;
    return X265GPU_OK;
}

void x265gpu_close(x265gpu_ctx* ctx)
{
    DeviceScope deviceScope(ctx);
    if (!ctx)
        return;
    (void)aq_finish_pending(ctx);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->slots)
    {
        if (s.base)
            (void)hipFree(s.base);
        if (s.full)
            (void)hipFree(s.full);
        if (s.aqStage)
            (void)hipHostFree(s.aqStage);
        if (s.aqStaged)
            (void)hipEventDestroy(s.aqStaged);
        for (char* w : s.weighted)
            if (w)
                (void)hipFree(w);
    }
    if (ctx->costBuf) (void)hipFree(ctx->costBuf);
    if (ctx->jobBuf) (void)hipFree(ctx->jobBuf);
    if (ctx->jobHost) (void)hipHostFree(ctx->jobHost);
    if (ctx->jobCopied) (void)hipEventDestroy(ctx->jobCopied);
    if (ctx->srcStage) (void)hipFree(ctx->srcStage);
    if (ctx->scalar) (void)hipFree(ctx->scalar);
    if (ctx->batchSums) (void)hipFree(ctx->batchSums);
    if (ctx->batchSumsHost) (void)hipHostFree(ctx->batchSumsHost);
    for (int k = 0; k < 3; k++)
    {
        if (ctx->searchAux.side[k]) (void)hipStreamSynchronize(ctx->searchAux.side[k]);
        if (ctx->searchAux.join[k]) (void)hipEventDestroy(ctx->searchAux.join[k]);
        if (ctx->searchAux.side[k]) (void)hipStreamDestroy(ctx->searchAux.side[k]);
    }
    if (ctx->searchAux.fork) (void)hipEventDestroy(ctx->searchAux.fork);
    if (ctx->meCostBuf) (void)hipFree(ctx->meCostBuf);
    if (ctx->puBuf) (void)hipFree(ctx->puBuf);
    if (ctx->mcstfBuf) (void)hipFree(ctx->mcstfBuf);
    if (ctx->seaBuf) (void)hipFree(ctx->seaBuf);
    if (ctx->intraStream) (void)hipStreamSynchronize(ctx->intraStream);
    if (ctx->aqDev) (void)hipFree(ctx->aqDev);
    if (ctx->motionDev) (void)hipFree(ctx->motionDev);
    if (ctx->motionHost) (void)hipHostFree(ctx->motionHost);
    if (ctx->cutreeBuf) (void)hipFree(ctx->cutreeBuf);
    if (ctx->aqHost) (void)hipHostFree(ctx->aqHost);
    if (ctx->aqInstalled) (void)hipEventDestroy(ctx->aqInstalled);
    if (ctx->finishHost) (void)hipHostFree(ctx->finishHost);
    if (ctx->intraJobBuf) (void)hipFree(ctx->intraJobBuf);
    if (ctx->intraJobHost) (void)hipHostFree(ctx->intraJobHost);
    if (ctx->intraFork) (void)hipEventDestroy(ctx->intraFork);
    if (ctx->intraDone) (void)hipEventDestroy(ctx->intraDone);
    if (ctx->intraStream) (void)hipStreamDestroy(ctx->intraStream);
    if (ctx->searchSync) (void)hipFree(ctx->searchSync);
    for (uint32_t* t : ctx->flowChunks)
        if (t) (void)hipFree(t);
    if (ctx->searchErrHost) (void)hipHostFree(ctx->searchErrHost);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    deviceScope.release();
    delete ctx;
}

int x265gpu_get_geometry(const x265gpu_ctx* ctx, x265gpu_geometry* o)
{
    if (!ctx || !o)
        return X265GPU_ERR_ARG;
    const Geometry& g = ctx->g;
    o->width = g.width;
    o->lines = g.lines;
    o->stride = g.stride;
    o->marginX = g.marginX;
    o->marginY = g.marginY;
    o->widthInCU = g.W;
    o->heightInCU = g.H;
    o->cuCount = g.cuCount;
    o->widthInCU4 = g.W4;
    o->heightInCU4 = g.H4;
    o->planeSize = g.planeSize;
    o->padOffset = g.padOffset;
    o->numCoopSlices = ctx->numCoopSlices;
    o->numRowsPerSlice = ctx->numRowsPerSlice;
    o->sourceWidth = ctx->cfg.sourceWidth;
    o->sourceHeight = ctx->cfg.sourceHeight;
    return X265GPU_OK;
}

const char* x265gpu_last_error(const x265gpu_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int x265gpu_host_alloc(x265gpu_ctx* ctx, size_t bytes, void** out)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !out || !bytes)
        return X265GPU_ERR_ARG;
    *out = nullptr;
    HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return X265GPU_OK;
}

int x265gpu_host_free(x265gpu_ctx* ctx, void* p)
{
    DeviceScope deviceScope(ctx);
    if (!ctx)
        return X265GPU_ERR_ARG;
    if (p)
        HIPCHK(hipHostFree(p));
    return X265GPU_OK;
}

int x265gpu_sync(x265gpu_ctx* ctx)
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);
    if (!ctx)
        return X265GPU_ERR_ARG;
    return verify_searches(ctx);
}

int x265gpu_set_mvcost(x265gpu_ctx* ctx, const uint16_t* centre, int32_t maxMvd)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !centre || maxMvd < 1024)
        return X265GPU_ERR_ARG;
    if (ctx->costBuf)
    {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(ctx->costBuf));
        ctx->costBuf = nullptr;
    }
    const size_t count = 2 * (size_t)maxMvd + 1;
    HIPCHK(hipMalloc((void**)&ctx->costBuf, count * sizeof(uint16_t)));
    HIPCHK(hipMemcpyAsync(ctx->costBuf, centre - maxMvd, count * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->costCentre = ctx->costBuf + maxMvd;
    ctx->maxMvd = maxMvd;
    return X265GPU_OK;
}

int x265gpu_lowres_init_batch(x265gpu_ctx* ctx, int32_t n, const int32_t* slots, const void* const* lumas,
                              const intptr_t* strides, int32_t memKind, const int32_t* pocs)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || n < 0 || (n && (!slots || !lumas || !strides)))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    const Geometry& g = ctx->g;
    for (int i = 0; i < n; i++)
        if (!slot_ok(ctx, slots[i]) || !lumas[i] || strides[i] < g.srcWidth)
            return fail(ctx, X265GPU_ERR_ARG, "lowres_init: bad slot, plane or stride");
    // cuTree passes queued so far hold the addresses of what this call rewrites: they run first
    { const int rcTree = cutree_flush(ctx); if (rcTree) return rcTree; }

    std::vector<LowresJob> jobs(n);
    const size_t frameBytes = (size_t)g.srcWidth * g.srcHeight * ctx->bpp;
    if (memKind == X265GPU_MEM_HOST)
    {
        int rc = ensure_src_stage(ctx, frameBytes * n);
        if (rc)
            return rc;
    }
    for (int i = 0; i < n; i++)
    {
        Slot& s = ctx->slots[slots[i]];
        s.poc = pocs ? pocs[i] : 0;
        s.intraFresh = false;
        if (memKind == X265GPU_MEM_HOST)
        {
            char* dst = ctx->srcStage + frameBytes * i;
            HIPCHK(hipMemcpy2DAsync(dst, (size_t)g.srcWidth * ctx->bpp, lumas[i], (size_t)strides[i] * ctx->bpp,
                                    (size_t)g.srcWidth * ctx->bpp, g.srcHeight, hipMemcpyHostToDevice, ctx->stream));
            jobs[i].src = dst;
            jobs[i].srcStride = g.srcWidth;
        }
        else
        {
            jobs[i].src = lumas[i];
            jobs[i].srcStride = strides[i];
        }
        jobs[i].planes = s.planes;
        jobs[i].planes4 = s.planes4;
        if (s.full)
        {
            HIPCHK(launch_pad_picture(ctx->stream, ctx->bpp, jobs[i].src, jobs[i].srcStride, s.full, ctx->fullStride,
                                      ctx->fullRows, g.srcWidth, g.srcHeight, ctx->fullMarginX, ctx->fullMarginY));
            s.fullSerial++;
        }
    }
    int rc = ensure_job_buf(ctx, sizeof(LowresJob) * n);
    if (rc)
        return rc;
    rc = upload_jobs(ctx, 0, jobs.data(), sizeof(LowresJob) * n, true);
    if (rc)
        return rc;
    PhaseTimer pt(ctx, X265GPU_PHASE_LOWRES);
    HIPCHK(launch_lowres_init(ctx->stream, g, ctx->bpp, (const LowresJob*)ctx->jobBuf, n));
    pt.stop(g.stride4 > 0 ? 2 : 1);
    return X265GPU_OK;
}

int x265gpu_lowres_init(x265gpu_ctx* ctx, int32_t slot, const void* luma, intptr_t stride, int32_t memKind,
                        int32_t poc)
{
    DeviceScope deviceScope(ctx);
    return x265gpu_lowres_init_batch(ctx, 1, &slot, &luma, &stride, memKind, &poc);
}

static int set_aq_staged(x265gpu_ctx* ctx, int32_t slot, const int32_t* invQscaleFactor, hipStream_t stream = nullptr,
                         bool redoIntraSums = true)
{
    if (!stream)
        stream = ctx->stream;
    Slot& s = ctx->slots[slot];
    // Factors that arrive after the picture's intra estimate: its AQ-weighted sums are redone (aq_finish_pending)
    // before anything reads them, behind this upload.
    if (s.intraFresh && redoIntraSums)
    {
        ctx->intraDeferred.push_back(slot);
        ctx->aqInstalledOnMain = true;
    }
    if (!invQscaleFactor)
    {
        s.hasAq = false;
        return X265GPU_OK;
    }
    // The upload reads a staging copy the slot owns, so the caller's array is theirs again on return whatever
    // kind of memory it is, and the copy still overlaps the stream's earlier work.
    const size_t bytes = sizeof(int32_t) * ctx->g.cuCount;
    if (!s.aqStage)
    {
        HIPCHK(hipHostMalloc((void**)&s.aqStage, bytes, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&s.aqStaged, hipEventDisableTiming));
    }
    else
        HIPCHK(hipEventSynchronize(s.aqStaged));      // the previous upload of this slot has left the staging copy
    memcpy(s.aqStage, invQscaleFactor, bytes);
    HIPCHK(hipMemcpyAsync(s.invQscale, s.aqStage, bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(hipEventRecord(s.aqStaged, stream));
    s.hasAq = true;
    return X265GPU_OK;
}

int x265gpu_lowres_set_aq(x265gpu_ctx* ctx, int32_t slot, const int32_t* invQscaleFactor)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slot))
        return X265GPU_ERR_ARG;
    int rc = aq_finish_pending(ctx);
    if (rc)
        return rc;
    // cuTree passes queued so far hold the addresses of what this call rewrites: they run first
    { const int rcTree = cutree_flush(ctx); if (rcTree) return rcTree; }
    return set_aq_staged(ctx, slot, invQscaleFactor);
}

int x265gpu_intra_estimate(x265gpu_ctx* ctx, int32_t n, const int32_t* slots)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || n < 0 || (n && !slots))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    for (int i = 0; i < n; i++)
        if (!slot_ok(ctx, slots[i]))
            return fail(ctx, X265GPU_ERR_ARG, "intra_estimate: bad slot");
    // cuTree passes queued so far hold the addresses of what this call rewrites: they run first
    { const int rcTree = cutree_flush(ctx); if (rcTree) return rcTree; }
    // Scale factors that host threads are still computing (x265gpu_aq_frames_begin) only enter the AQ-weighted sums:
    // the estimate itself starts now, so that what reads its per-block costs (weight analysis, cost passes) finds
    // them, and the sums are redone when the factors are installed (aq_finish_pending), ahead of whatever reads those.
    const bool aqLater = !ctx->aqPending.empty();
    if (aqLater)
        ctx->intraDeferred.insert(ctx->intraDeferred.end(), slots, slots + n);
    std::vector<IntraJob> jobs(n);
    for (int i = 0; i < n; i++)
    {
        Slot& s = ctx->slots[slots[i]];
        jobs[i].plane0 = plane_origin(ctx, s.planes);
        jobs[i].invQscale = (s.hasAq && !aqLater) ? s.invQscale : nullptr;
        s.intraFresh = true;
        jobs[i].intraCost = s.intraCost;
        jobs[i].intraMode = s.intraMode;
        jobs[i].lowresCosts = s.lowresCosts;   // [0][0]
        jobs[i].rowSatds = s.rowSatds;         // [0][0]
        jobs[i].rowPartials = s.rowPartials;   // scratch of request (0,0)
        jobs[i].sums = s.sums;                 // totals of request (0,0)
    }
    if (ctx->intraStream && !ctx->prof)
    {
        // side stream: starts after everything queued on the main stream so far (planes, AQ factors)
        const size_t bytes = sizeof(IntraJob) * n;
        int rcBuf = ensure_intra_job_buf(ctx, bytes);       // waits until the previous launch is done with the buffers
        if (rcBuf)
            return rcBuf;
        memcpy(ctx->intraJobHost, jobs.data(), bytes);
        if (!ctx->intraNoFork)
        {
            HIPCHK(hipEventRecord(ctx->intraFork, ctx->stream));
            HIPCHK(hipStreamWaitEvent(ctx->intraStream, ctx->intraFork, 0));
        }
        HIPCHK(hipMemcpyAsync(ctx->intraJobBuf, ctx->intraJobHost, bytes, hipMemcpyHostToDevice, ctx->intraStream));
        HIPCHK(launch_intra(ctx->intraStream, ctx->g, ctx->bpp, ctx->cfg.internalBitDepth, ctx->cfg.lookaheadLambda,
                            (const IntraJob*)ctx->intraJobBuf, n));
        HIPCHK(hipEventRecord(ctx->intraDone, ctx->intraStream));
        ctx->intraPending = true;
        return X265GPU_OK;
    }
    int rc = ensure_job_buf(ctx, sizeof(IntraJob) * n);
    if (rc)
        return rc;
    rc = upload_jobs(ctx, 0, jobs.data(), sizeof(IntraJob) * n, true);
    if (rc)
        return rc;
    PhaseTimer pt(ctx, X265GPU_PHASE_INTRA);
    HIPCHK(launch_intra(ctx->stream, ctx->g, ctx->bpp, ctx->cfg.internalBitDepth, ctx->cfg.lookaheadLambda,
                        (const IntraJob*)ctx->jobBuf, n));
    pt.stop(2);
    return X265GPU_OK;
}

static int run_estimate(x265gpu_ctx* ctx, int32_t n, const x265gpu_request* reqs, bool wantBatchOut,
                        long long* deviceOut)
{
    if (!ctx || n < 0 || (n && !reqs))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    const Geometry& g = ctx->g;
    if (wantBatchOut && (size_t)n > ctx->batchSumsCap)
    {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->batchSums) HIPCHK(hipFree(ctx->batchSums));
        if (ctx->batchSumsHost) HIPCHK(hipHostFree(ctx->batchSumsHost));
        ctx->batchSums = nullptr;
        ctx->batchSumsHost = nullptr;
        ctx->batchSumsCap = 0;
        const size_t cap = (size_t)n * 2;
        HIPCHK(hipMalloc((void**)&ctx->batchSums, cap * 3 * sizeof(long long)));
        HIPCHK(hipHostMalloc((void**)&ctx->batchSumsHost, cap * 3 * sizeof(long long), hipHostMallocDefault));
        ctx->batchSumsCap = cap;
    }
    const int D = ctx->D, nb = g.cuCount, nb4 = g.cuCount4;
    const bool hme = ctx->cfg.bEnableHME != 0;
    std::vector<SearchJob> sj;      // half-res searches
    std::vector<SearchJob> sj4;     // quarter-res searches (HME level 0), run first
    std::vector<CostJob> cj;
    cj.reserve(n);
    int slicesMax = 1, sliceHeightMax = 0;
    for (int i = 0; i < n; i++)
    {
        const x265gpu_request& r = reqs[i];
        const bool coop = r.bCoopSlices && ctx->numCoopSlices > 1;
        // x265 runs both levels of a slice in one task and a slice's half-res blocks read quarter-res
        // vectors of rows another slice's task writes (slicetype.cpp:4250-4276, :4450-4453): the
        // result depends on thread timing there, so there is nothing to be exact against
        if (coop && hme && (r.doSearch[0] || r.doSearch[1]))
            return fail(ctx, X265GPU_ERR_UNSUPPORTED, "estimate: cooperative slices with HME");
        const int sliceHeight = coop ? g.H - ctx->numRowsPerSlice * (ctx->numCoopSlices - 1) : g.H;
        if (r.doSearch[0] || r.doSearch[1])
        {
            slicesMax = coop && ctx->numCoopSlices > slicesMax ? ctx->numCoopSlices : slicesMax;
            sliceHeightMax = sliceHeight > sliceHeightMax ? sliceHeight : sliceHeightMax;
        }
        const int d0 = r.b - r.p0, d1 = r.p1 - r.b;
        // d0 == 0 with d1 > 0 is a request x265 does issue: a leading B frame after a closed-GOP keyframe has no
        // past reference and is measured with p0 = b (slicetype.cpp:2647-2660), i.e. against itself in list 0
        if (!slot_ok(ctx, r.slotP0) || !slot_ok(ctx, r.slotB) || !slot_ok(ctx, r.slotP1) || d0 < 0 || d0 >= D ||
            d1 < 0 || d1 >= D || (d0 == 0 && d1 == 0) || (d0 == 0 && r.slotP0 != r.slotB) ||
            (d1 == 0 && r.slotP1 != r.slotB) || (r.doSearch[1] && d1 == 0))
            return fail(ctx, X265GPU_ERR_ARG, "estimate: bad request");
        if ((r.doSearch[0] || r.doSearch[1]) && !ctx->costCentre)
            return fail(ctx, X265GPU_ERR_STATE, "estimate: MV cost table not set");
        Slot& sb = ctx->slots[r.slotB];
        Slot& s0 = ctx->slots[r.slotP0];
        Slot& s1 = ctx->slots[r.slotP1];
        const int bBidir = d1 > 0;
        int32_t* mvs[2] = { sb.mvs + ((size_t)(0 * D + d0) * nb) * 2, sb.mvs + ((size_t)(1 * D + d1) * nb) * 2 };
        int32_t* mc[2] = { sb.mvCosts + (size_t)(0 * D + d0) * nb, sb.mvCosts + (size_t)(1 * D + d1) * nb };
        for (int l = 0; l < 2; l++)
        {
            if (!r.doSearch[l])
                continue;
            Slot& sr = l ? s1 : s0;
            const int dist = l ? d1 : d0;
            int32_t* mvs4 = hme ? sb.mvs4 + ((size_t)(l * D + dist) * nb4) * 2 : nullptr;
            int32_t* mc4 = hme ? sb.mvCosts4 + (size_t)(l * D + dist) * nb4 : nullptr;
            if (hme)
            {
                // estimateCUCost(..., hme = 1): quarter-res planes, never the weighted reference
                // (slicetype.cpp:4391); the closing zero check still addresses the half-res plane
                // with the full stride (motion.cpp:1765, 1777-1778)
                SearchJob j;
                memset(&j, 0, sizeof(j));
                j.fencPlane0 = plane4_origin(ctx, sb.planes4);
                j.refOrigin = plane4_origin(ctx, sr.planes4);
                j.zeroRef = plane_origin(ctx, sr.planes);
                j.zeroStride = g.stride;
                j.mvs = mvs4;
                j.mvCosts = mc4;
                j.bBidir = bBidir;
                j.level0 = 1;
                j.method = ctx->cfg.hmeSearchMethod[0];
                j.merange = ctx->cfg.hmeRange[0];
                j.numSlices = 1;
                j.sliceRows = g.H4;
                sj4.push_back(j);
            }
            // Searches of one call run side by side on several streams and read their own earlier vectors, so two of them
            // must never share a (frame, list, distance) vector array: the second would read what the first is writing.
            for (const SearchJob& other : sj)
                if (other.mvs == mvs[l])
                    return fail(ctx, X265GPU_ERR_ARG, "estimate: two searches of the same frame, list and distance in one call");
            SearchJob j;
            memset(&j, 0, sizeof(j));
            j.fencPlane0 = plane_origin(ctx, sb.planes);
            char* ref = sr.planes;
            if (l == 0 && r.weightedL0)
            {
                if (!sb.weighted[d0])
                    return fail(ctx, X265GPU_ERR_STATE, "estimate: weighted planes not prepared");
                ref = sb.weighted[d0];
            }
            j.refOrigin = plane_origin(ctx, ref);
            j.zeroRef = j.refOrigin;
            j.zeroStride = g.stride;
            j.mvs = mvs[l];
            j.mvCosts = mc[l];
            j.hmeMvs = mvs4;
            j.hmeMvCosts = mc4;
            j.bBidir = bBidir;
            // with HME the half-res level uses hmeSearchMethod[1] / hmeRange[1] (motion.cpp:1013,
            // slicetype.cpp:4478); without it x265 hard-wires HEX and s_merange
            j.method = hme ? ctx->cfg.hmeSearchMethod[1] : ctx->cfg.searchMethod;
            j.merange = hme ? ctx->cfg.hmeRange[1] : ctx->cfg.merange;
            j.numSlices = coop ? ctx->numCoopSlices : 1;
            j.sliceRows = coop ? ctx->numRowsPerSlice : g.H;
            sj.push_back(j);
        }
        if (r.bSearchOnly)
            continue;
        cj.emplace_back();
        CostJob& c = cj.back();
        memset(&c, 0, sizeof(c));
        const size_t q = (size_t)d0 * D + d1;
        c.fencPlane0 = plane_origin(ctx, sb.planes);
        c.ref0Origin = plane_origin(ctx, s0.planes);
        c.ref1Origin = plane_origin(ctx, s1.planes);
        c.mvs0 = mvs[0];
        c.mvs1 = mvs[1];
        c.mvCosts0 = mc[0];
        c.mvCosts1 = mc[1];
        c.intraCost = sb.intraCost;
        c.invQscale = sb.hasAq ? sb.invQscale : nullptr;
        c.lowresCosts = sb.lowresCosts + q * nb;
        c.rowSatds = sb.rowSatds + q * g.H;
        c.rowPartials = sb.rowPartials + q * g.H * 3;
        c.sums = sb.sums + q * 3;
        c.batchOut = deviceOut ? deviceOut + (size_t)i * 3 : wantBatchOut ? ctx->batchSums + (size_t)i * 3 : nullptr;
        c.bBidir = bBidir;
    }

    const size_t s4Bytes = align256(sizeof(SearchJob) * sj4.size());
    const size_t sBytes = align256(sizeof(SearchJob) * sj.size());
    const size_t cBytes = sizeof(CostJob) * cj.size();
    int rc = ensure_job_buf(ctx, s4Bytes + sBytes + cBytes);
    if (rc)
        return rc;
    bool first = true;
    if (!sj4.empty())
    {
        rc = upload_jobs(ctx, 0, sj4.data(), sizeof(SearchJob) * sj4.size(), first);
        if (rc)
            return rc;
        first = false;
    }
    if (!sj.empty())
    {
        rc = upload_jobs(ctx, s4Bytes, sj.data(), sizeof(SearchJob) * sj.size(), first);
        if (rc)
            return rc;
        first = false;
    }
    if (!cj.empty())
    {
        rc = upload_jobs(ctx, s4Bytes + sBytes, cj.data(), cBytes, first);
        if (rc)
            return rc;
    }

    const bool singleLaunch = !sj.empty() && !ctx->singleLaunchOff;
    if (singleLaunch && (ctx->searchRows || ctx->searchFlow))
    {
        // kept until a call that waits has seen the error word clear (verify_searches)
        if (ctx->unverified.size() >= 256)
            ctx->unverified.erase(ctx->unverified.begin());
        ctx->unverified.push_back({ std::vector<x265gpu_request>(reqs, reqs + n), wantBatchOut, deviceOut });
    }
    if (singleLaunch && ctx->searchRows)
    {
        rc = ensure_search_sync(ctx, (int)sj.size());
        if (rc)
            return rc;
        PhaseTimer pt(ctx, X265GPU_PHASE_SEARCH);
        int launches = 0;
        if (!sj4.empty())
            HIPCHK(launch_search_rows(ctx->stream, g, ctx->bpp, (const SearchJob*)ctx->jobBuf, (int)sj4.size(), 1,
                                      sj4[0].method, ctx->costCentre, ctx->maxMvd, ctx->searchSync, &launches));
        HIPCHK(launch_search_rows(ctx->stream, g, ctx->bpp, (const SearchJob*)(ctx->jobBuf + s4Bytes), (int)sj.size(), 0,
                                  sj[0].method, ctx->costCentre, ctx->maxMvd, ctx->searchSync, &launches));
        pt.stop(launches);
    }
    else if (singleLaunch && ctx->searchFlow && slicesMax == 1 && ctx->flowChunkCount[0] <= 65535 &&
             (sj4.empty() || ctx->flowChunkCount[1] <= 65535))
    {
        rc = ensure_search_sync(ctx, 1);
        if (rc)
            return rc;
        if (ctx->flowFaultAfter >= 0 && ctx->flowLaunches++ == ctx->flowFaultAfter)
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(ctx->searchSync + SEARCH_SYNC_ERROR), 1, 1, ctx->stream));
        PhaseTimer pt(ctx, X265GPU_PHASE_SEARCH);
        int launches = 0;
        if (!sj4.empty())
            HIPCHK(launch_search_flow(ctx->stream, g, ctx->bpp, (const SearchJob*)ctx->jobBuf, (int)sj4.size(), 1, sj4[0].method,
                                      ctx->flowChunks[1], ctx->flowChunkCount[1], ctx->costCentre, ctx->maxMvd, ctx->searchSync,
                                      ctx->flowGridPad, &launches));
        HIPCHK(launch_search_flow(ctx->stream, g, ctx->bpp, (const SearchJob*)(ctx->jobBuf + s4Bytes), (int)sj.size(), 0,
                                  sj[0].method, ctx->flowChunks[0], ctx->flowChunkCount[0], ctx->costCentre, ctx->maxMvd,
                                  ctx->searchSync, ctx->flowGridPad, &launches));
        pt.stop(launches);
    }
    else if (!sj.empty())
    {
        PhaseTimer pt(ctx, X265GPU_PHASE_SEARCH);
        int launches = 0;
        if (!sj4.empty())
            HIPCHK(launch_search(ctx->stream, ctx->searchAux, g, ctx->bpp, (const SearchJob*)ctx->jobBuf, (int)sj4.size(), 1,
                                 sj4[0].method, ctx->costCentre, ctx->maxMvd, 1, g.H4, &launches));
        HIPCHK(launch_search(ctx->stream, ctx->searchAux, g, ctx->bpp, (const SearchJob*)(ctx->jobBuf + s4Bytes), (int)sj.size(), 0,
                             sj[0].method, ctx->costCentre, ctx->maxMvd, slicesMax, sliceHeightMax, &launches));
        pt.stop(launches);
    }
    if (!cj.empty())
    {
        // scale factors (and the intra estimates that wait for them) of x265gpu_aq_frames_begin batches: the host's
        // double-precision mapping ran beside the searches queued above
        rc = aq_finish_pending(ctx);
        if (rc)
            return rc;
    }
    join_intra(ctx);          // the cost pass reads intraCost; the searches above did not have to wait for it
    if (!cj.empty())
    {
        PhaseTimer pt(ctx, X265GPU_PHASE_COST);
        HIPCHK(launch_cost(ctx->stream, g, ctx->bpp, (const CostJob*)(ctx->jobBuf + s4Bytes + sBytes), (int)cj.size()));
        pt.stop(2);
    }
    return X265GPU_OK;
}

int x265gpu_estimate(x265gpu_ctx* ctx, int32_t n, const x265gpu_request* reqs)
{
    DeviceScope deviceScope(ctx, false);
    return run_estimate(ctx, n, reqs, false, nullptr);
}

int x265gpu_estimate_sums_device(x265gpu_ctx* ctx, int32_t n, const x265gpu_request* reqs, int64_t* deviceSums)
{
    DeviceScope deviceScope(ctx, false);
    if (!deviceSums && n)
        return X265GPU_ERR_ARG;
    static_assert(sizeof(long long) == sizeof(int64_t), "frame totals are 64-bit");
    return run_estimate(ctx, n, reqs, false, (long long*)deviceSums);
}

int x265gpu_estimate_costs(x265gpu_ctx* ctx, int32_t n, const x265gpu_request* reqs, x265gpu_frame_cost* out)
{
    DeviceScope deviceScope(ctx, false);
    if (!out && n)
        return X265GPU_ERR_ARG;
    int rc = run_estimate(ctx, n, reqs, true, nullptr);
    if (rc || !n)
        return rc;
    HIPCHK(hipMemcpyAsync(ctx->batchSumsHost, ctx->batchSums, (size_t)n * 3 * sizeof(long long),
                          hipMemcpyDeviceToHost, ctx->stream));
    bool recovered = false;
    rc = verify_searches(ctx, &recovered);
    if (rc)
        return rc;
    if (recovered)
    {
        // the batch was run again: its totals once more
        HIPCHK(hipMemcpyAsync(ctx->batchSumsHost, ctx->batchSums, (size_t)n * 3 * sizeof(long long),
                              hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    for (int i = 0; i < n; i++)
    {
        if (reqs[i].bSearchOnly)
        {
            out[i].costEst = out[i].costEstAq = -1;
            out[i].intraMbs = out[i].reserved = 0;
            continue;
        }
        long long score = ctx->batchSumsHost[3 * i];
        if (reqs[i].p1 != reqs[i].b)
            score = score * 100 / (130 + ctx->cfg.bFrameBias);   // slicetype.cpp:4372-4373
        out[i].costEst = score;
        out[i].costEstAq = ctx->batchSumsHost[3 * i + 1];
        out[i].intraMbs = (int32_t)ctx->batchSumsHost[3 * i + 2];
        out[i].reserved = 0;
    }
    return X265GPU_OK;
}

int x265gpu_weight_cost(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef, int32_t wtPresent, int32_t scale,
                        int32_t log2Denom, int32_t offset, uint32_t* cost)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slotFenc) || !slot_ok(ctx, slotRef) || !cost)
        return X265GPU_ERR_ARG;
    const Geometry& g = ctx->g;
    Slot& sf = ctx->slots[slotFenc];
    Slot& sr = ctx->slots[slotRef];
    const int depth = ctx->cfg.internalBitDepth;
    WeightCostJob job{};
    job.fencPlane0 = plane_origin(ctx, sf.planes);
    job.refPlane0 = plane_origin(ctx, sr.planes);
    job.intraCost = sf.intraCost;
    job.out = ctx->scalar;
    if (wtPresent)
    {
        // weight plane 0 of the reference into the staging buffer (slicetype.cpp:822-834)
        int rc = ensure_src_stage(ctx, (size_t)g.planeSize * ctx->bpp);
        if (rc)
            return rc;
        const int corr = 14 - depth;
        const int round = log2Denom ? 1 << (log2Denom - 1) : 0;
        HIPCHK(launch_weight_planes(ctx->stream, g, ctx->bpp, depth, 1, sr.planes, ctx->srcStage, scale,
                                    round << corr, log2Denom + corr, offset * (1 << (depth - 8))));
        job.refPlane0 = plane_origin(ctx, ctx->srcStage);
    }
    HIPCHK(hipMemsetAsync(ctx->scalar, 0, sizeof(unsigned int), ctx->stream));
    HIPCHK(launch_weight_cost(ctx->stream, g, ctx->bpp, &job));
    HIPCHK(hipMemcpyAsync(cost, ctx->scalar, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

namespace {
int ensure_pu_buf(x265gpu_ctx* ctx, size_t bytes);
}

int x265gpu_weight_costs(x265gpu_ctx* ctx, int32_t n, const x265gpu_weight_job* jobs, uint32_t* costs)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || n < 0 || (n && (!jobs || !costs)))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    const Geometry& g = ctx->g;
    const int depth = ctx->cfg.internalBitDepth;
    for (int i = 0; i < n; i++)
        if (!slot_ok(ctx, jobs[i].slotFenc) || !slot_ok(ctx, jobs[i].slotRef))
            return fail(ctx, X265GPU_ERR_ARG, "weight_costs: bad slot");
    const size_t accBytes = align256(sizeof(unsigned int) * (size_t)n);
    int rc = ensure_pu_buf(ctx, accBytes + sizeof(WeightCostJob) * (size_t)n);
    if (rc)
        return rc;
    unsigned int* acc = (unsigned int*)ctx->puBuf;
    HIPCHK(hipMemsetAsync(acc, 0, sizeof(unsigned int) * (size_t)n, ctx->stream));
    std::vector<WeightCostJob> batch((size_t)n);
    for (int i = 0; i < n; i++)
    {
        const x265gpu_weight_job& j = jobs[i];
        Slot& sf = ctx->slots[j.slotFenc];
        Slot& sr = ctx->slots[j.slotRef];
        WeightCostJob& job = batch[i];
        memset(&job, 0, sizeof(job));
        job.fencPlane0 = plane_origin(ctx, sf.planes);
        job.refPlane0 = plane_origin(ctx, sr.planes);
        job.intraCost = sf.intraCost;
        job.out = acc + i;
        if (j.wtPresent)
        {
            // the weighted reference (slicetype.cpp:822-834) is never written out: the kernel weights each row as
            // it reads it
            const int corr = 14 - depth;
            const int round = j.log2Denom ? 1 << (j.log2Denom - 1) : 0;
            job.weighted = 1;
            job.w0 = j.scale;
            job.round = round << corr;
            job.shift = j.log2Denom + corr;
            job.offset = j.offset * (1 << (depth - 8));
            job.correction = corr;
            job.pixelMax = (1 << depth) - 1;
        }
    }
    // one launch for every pair
    WeightCostJob* dBatch = (WeightCostJob*)(ctx->puBuf + accBytes);
    HIPCHK(hipMemcpyAsync(dBatch, batch.data(), sizeof(WeightCostJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(launch_weight_cost_batch(ctx->stream, g, ctx->bpp, dBatch, n));
    HIPCHK(hipMemcpyAsync(costs, acc, sizeof(unsigned int) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_weight_cost_static(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef, int32_t wtPresent, int32_t scale,
                               int32_t log2Denom, int32_t offset, uint32_t* cost)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slotFenc) || !slot_ok(ctx, slotRef) || !cost)
        return X265GPU_ERR_ARG;
    if (!ctx->staticScratch)
        return fail(ctx, X265GPU_ERR_STATE, "weight_cost_static: no static scratch on this device");
    // the call waits for its result before it returns, so holding the device's lock until then keeps calls of
    // different contexts on one GPU out of each other's accumulator and plane
    std::lock_guard<std::mutex> scratchLock(g_scratchMutex[ctx->cfg.deviceId]);
    const Geometry& g = ctx->g;
    const size_t planeBytes = (size_t)g.planeSize * ctx->bpp;
    if (wtPresent && STATIC_SCRATCH_PLANE + planeBytes > STATIC_SCRATCH_BYTES)
    {
        char msg[160];
        snprintf(msg, sizeof(msg), "weight_cost_static: a padded plane of %zu bytes does not fit the %zu-byte static scratch",
                 planeBytes, STATIC_SCRATCH_BYTES - STATIC_SCRATCH_PLANE);
        return fail(ctx, X265GPU_ERR_ARG, msg);
    }
    Slot& sf = ctx->slots[slotFenc];
    Slot& sr = ctx->slots[slotRef];
    const int depth = ctx->cfg.internalBitDepth;
    unsigned int* acc = (unsigned int*)ctx->staticScratch;
    char* plane = ctx->staticScratch + STATIC_SCRATCH_PLANE;
    WeightCostJob job{};
    job.fencPlane0 = plane_origin(ctx, sf.planes);
    job.refPlane0 = plane_origin(ctx, sr.planes);
    job.intraCost = sf.intraCost;
    job.out = acc;
    if (wtPresent)
    {
        const int corr = 14 - depth;
        const int round = log2Denom ? 1 << (log2Denom - 1) : 0;
        HIPCHK(launch_weight_planes(ctx->stream, g, ctx->bpp, depth, 1, sr.planes, plane, scale, round << corr,
                                    log2Denom + corr, offset * (1 << (depth - 8))));
        job.refPlane0 = plane_origin(ctx, plane);
    }
    // the array keeps the previous call's contents: the accumulator must start at zero (the plane is
    // overwritten in full by the pass above before it is read)
    HIPCHK(hipMemsetAsync(acc, 0, sizeof(unsigned int), ctx->stream));
    HIPCHK(launch_weight_cost(ctx->stream, g, ctx->bpp, &job));
    HIPCHK(hipMemcpyAsync(cost, acc, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_weight_planes(x265gpu_ctx* ctx, int32_t slotFenc, int32_t dist, int32_t slotRef, int32_t scale,
                          int32_t log2Denom, int32_t offset)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slotFenc) || !slot_ok(ctx, slotRef) || dist < 1 || dist >= ctx->D)
        return X265GPU_ERR_ARG;
    const Geometry& g = ctx->g;
    Slot& sf = ctx->slots[slotFenc];
    Slot& sr = ctx->slots[slotRef];
    if (!sf.weighted[dist])
        HIPCHK(hipMalloc((void**)&sf.weighted[dist], (size_t)4 * g.planeSize * ctx->bpp));
    const int depth = ctx->cfg.internalBitDepth;
    const int corr = 14 - depth;
    const int round = log2Denom ? 1 << (log2Denom - 1) : 0;
    HIPCHK(launch_weight_planes(ctx->stream, g, ctx->bpp, depth, 4, sr.planes, sf.weighted[dist], scale, round << corr,
                                log2Denom + corr, offset * (1 << (depth - 8))));
    return X265GPU_OK;
}

int x265gpu_get_plane(x265gpu_ctx* ctx, int32_t slot, int32_t planeIdx, void* paddedBuffer)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slot) || planeIdx < 0 || planeIdx > 7 || !paddedBuffer)
        return X265GPU_ERR_ARG;
    const Geometry& g = ctx->g;
    Slot& s = ctx->slots[slot];
    if (planeIdx < 4)
        HIPCHK(hipMemcpyAsync(paddedBuffer, s.planes + (size_t)planeIdx * g.planeSize * ctx->bpp,
                              (size_t)g.planeSize * ctx->bpp, hipMemcpyDeviceToHost, ctx->stream));
    else
    {
        if (!s.planes4)
            return X265GPU_ERR_ARG;
        HIPCHK(hipMemcpyAsync(paddedBuffer, s.planes4 + (size_t)(planeIdx - 4) * g.planeSize4 * ctx->bpp,
                              (size_t)g.planeSize4 * ctx->bpp, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_get_intra(x265gpu_ctx* ctx, int32_t slot, int32_t* intraCost, uint8_t* intraMode)
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);
    if (!ctx || !slot_ok(ctx, slot))
        return X265GPU_ERR_ARG;
    Slot& s = ctx->slots[slot];
    if (intraCost)
        HIPCHK(hipMemcpyAsync(intraCost, s.intraCost, sizeof(int32_t) * ctx->g.cuCount, hipMemcpyDeviceToHost, ctx->stream));
    if (intraMode)
        HIPCHK(hipMemcpyAsync(intraMode, s.intraMode, ctx->g.cuCount, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_get_mvs(x265gpu_ctx* ctx, int32_t slot, int32_t list, int32_t dist, int32_t* mvsXY, int32_t* mvCosts)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slot) || list < 0 || list > 1 || dist < 0 || dist >= ctx->D)
        return X265GPU_ERR_ARG;
    Slot& s = ctx->slots[slot];
    const size_t nb = ctx->g.cuCount, idx = (size_t)list * ctx->D + dist;
    if (mvsXY)
        HIPCHK(hipMemcpyAsync(mvsXY, s.mvs + idx * nb * 2, sizeof(int32_t) * 2 * nb, hipMemcpyDeviceToHost, ctx->stream));
    if (mvCosts)
        HIPCHK(hipMemcpyAsync(mvCosts, s.mvCosts + idx * nb, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_get_hme_mvs(x265gpu_ctx* ctx, int32_t slot, int32_t list, int32_t dist, int32_t* mvsXY, int32_t* mvCosts)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slot) || list < 0 || list > 1 || dist < 0 || dist >= ctx->D || !ctx->g.cuCount4)
        return X265GPU_ERR_ARG;
    Slot& s = ctx->slots[slot];
    const size_t nb = ctx->g.cuCount4, idx = (size_t)list * ctx->D + dist;
    if (mvsXY)
        HIPCHK(hipMemcpyAsync(mvsXY, s.mvs4 + idx * nb * 2, sizeof(int32_t) * 2 * nb, hipMemcpyDeviceToHost, ctx->stream));
    if (mvCosts)
        HIPCHK(hipMemcpyAsync(mvCosts, s.mvCosts4 + idx * nb, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_get_costs(x265gpu_ctx* ctx, int32_t slot, int32_t d0, int32_t d1, uint16_t* lowresCosts,
                      int32_t* rowSatds, x265gpu_frame_cost* fc)
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);
    if (!ctx || !slot_ok(ctx, slot) || d0 < 0 || d0 >= ctx->D || d1 < 0 || d1 >= ctx->D)
        return X265GPU_ERR_ARG;
    Slot& s = ctx->slots[slot];
    const Geometry& g = ctx->g;
    const size_t q = (size_t)d0 * ctx->D + d1;
    long long sums[3] = { 0, 0, 0 };
    if (lowresCosts)
        HIPCHK(hipMemcpyAsync(lowresCosts, s.lowresCosts + q * g.cuCount, sizeof(uint16_t) * g.cuCount,
                              hipMemcpyDeviceToHost, ctx->stream));
    if (rowSatds)
        HIPCHK(hipMemcpyAsync(rowSatds, s.rowSatds + q * g.H, sizeof(int32_t) * g.H, hipMemcpyDeviceToHost, ctx->stream));
    if (fc)
        HIPCHK(hipMemcpyAsync(sums, s.sums + q * 3, sizeof(sums), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (fc)
    {
        long long score = sums[0];
        // B-frame bias, source/encoder/slicetype.cpp:4372-4373
        if (d1 != 0)
            score = score * 100 / (130 + ctx->cfg.bFrameBias);
        fc->costEst = score;
        fc->costEstAq = sums[1];
        fc->intraMbs = (d0 == 0 && d1 == 0) ? 0 : (int32_t)sums[2];
        fc->reserved = 0;
    }
    return X265GPU_OK;
}

namespace {
PuFrame pu_frame(const x265gpu_ctx* ctx, int slotFenc, int slotRef)
{
    PuFrame f;
    memset(&f, 0, sizeof(f));
    const size_t origin = ((size_t)ctx->fullMarginY * ctx->fullStride + ctx->fullMarginX) * ctx->bpp;
    f.fenc = ctx->slots[slotFenc].full + origin;
    f.ref = ctx->slots[slotRef].full + origin;
    f.stride = ctx->fullStride;
    f.width = ctx->fullW;
    f.height = ctx->fullH;
    f.marginX = ctx->fullMarginX;
    f.marginY = ctx->fullMarginY;
    f.depth = ctx->cfg.internalBitDepth;
    f.maxPasses = 1;
    static const bool rowForm = !getenv("X265GPU_PU_ROWFORM") || atoi(getenv("X265GPU_PU_ROWFORM")) != 0;
    f.rowForm = rowForm;
    return f;
}

int ensure_pu_buf(x265gpu_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->puBufSize)
        return X265GPU_OK;
    if (ctx->puBuf)
    {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(ctx->puBuf));
        ctx->puBuf = nullptr;
        ctx->puBufSize = 0;
    }
    HIPCHK(hipMalloc((void**)&ctx->puBuf, bytes));
    ctx->puBufSize = bytes;
    return X265GPU_OK;
}
} // namespace

int x265gpu_me_set_mvcost_tables(x265gpu_ctx* ctx, int32_t numTables, const uint16_t* const* centres, int32_t maxMvd)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !centres || numTables < 1 || numTables > X265GPU_PU_MAX_COST_TABLES || maxMvd < 1024)
        return X265GPU_ERR_ARG;
    for (int k = 0; k < numTables; k++)
        if (!centres[k])
            return X265GPU_ERR_ARG;
    if (ctx->meCostBuf)
    {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(ctx->meCostBuf));
        ctx->meCostBuf = nullptr;
    }
    // the tables behind each other, each 2 * maxMvd + 2 entries apart (an even count keeps them dword aligned)
    const size_t count = 2 * (size_t)maxMvd + 1, stride = count + 1;
    HIPCHK(hipMalloc((void**)&ctx->meCostBuf, numTables * stride * sizeof(uint16_t)));
    for (int k = 0; k < numTables; k++)
        HIPCHK(hipMemcpyAsync(ctx->meCostBuf + k * stride, centres[k] - maxMvd, count * sizeof(uint16_t), hipMemcpyHostToDevice,
                              ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->meCostCentre = ctx->meCostBuf + maxMvd;
    ctx->meMaxMvd = maxMvd;
    ctx->meCostTables = numTables;
    return X265GPU_OK;
}

int x265gpu_me_set_mvcost(x265gpu_ctx* ctx, const uint16_t* centre, int32_t maxMvd)
{
    return x265gpu_me_set_mvcost_tables(ctx, 1, &centre, maxMvd);
}

int x265gpu_pu_motion_estimate(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef, int32_t searchMethod, int32_t merange,
                               int32_t subpelRefine, int32_t n, const x265gpu_pu_job* jobs, x265gpu_pu_result* out)
{
    DeviceScope deviceScope(ctx);
    static_assert(sizeof(x265gpu_pu_job) == sizeof(PuJob) && sizeof(x265gpu_pu_result) == sizeof(PuResult),
                  "public and device job layouts must match");
    if (!ctx || n < 0 || (n && (!jobs || !out)))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    if (!slot_ok(ctx, slotFenc) || !slot_ok(ctx, slotRef))
        return fail(ctx, X265GPU_ERR_ARG, "pu_motion_estimate: bad slot");
    if (!ctx->fullStride)
        return fail(ctx, X265GPU_ERR_STATE, "pu_motion_estimate: context opened without bKeepFullRes");
    if (!ctx->meCostCentre)
        return fail(ctx, X265GPU_ERR_STATE, "pu_motion_estimate: MV cost table not set");
    if (searchMethod != X265GPU_DIA_SEARCH && searchMethod != X265GPU_HEX_SEARCH && searchMethod != X265GPU_UMH_SEARCH &&
        searchMethod != X265GPU_STAR_SEARCH && searchMethod != X265GPU_SEA_SEARCH &&
        searchMethod != X265GPU_FULL_SEARCH && searchMethod != X265GPU_AREA_DIAMOND_SEARCH)
        return fail(ctx, X265GPU_ERR_UNSUPPORTED, "pu_motion_estimate: search method");
    if (subpelRefine < 0 || subpelRefine > 7 || merange < 1 || merange > 32767)
        return fail(ctx, X265GPU_ERR_ARG, "pu_motion_estimate: subpelRefine or merange");
    const bool sea = searchMethod == X265GPU_SEA_SEARCH;
    if (sea && merange > 126)      // a row of the window is held as four 64-bit survivor masks
        return fail(ctx, X265GPU_ERR_UNSUPPORTED, "pu_motion_estimate: successive elimination with merange above 126");
    unsigned seaNeed = 0;
    // the jobs go up while the host checks and sorts them
    const size_t jobBytes = align256(sizeof(PuJob) * (size_t)n), outBytes = align256(sizeof(PuResult) * (size_t)n);
    const size_t orderBytes = sizeof(int) * (size_t)n;
    int rc = ensure_pu_buf(ctx, jobBytes + outBytes + orderBytes);
    if (rc)
        return rc;
    HIPCHK(hipMemcpyAsync(ctx->puBuf, jobs, sizeof(PuJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    int maxPasses = 1, midPasses = 1;
    // size classes: PUs of at most 2 / 4 / 8 tiles share a wave (8 / 4 / 2 at a time), the rest own one each, split
    // into those of at most 32 tiles (two passes, a small LDS record) and the large ones, which are started first,
    // largest first; X265GPU_PU_PACK=0 gives every PU a wave
    static const bool pack = []() { const char* e = getenv("X265GPU_PU_PACK"); return !e || atoi(e) != 0; }();
    std::vector<int> cls(n), passesOf(n);
    PuClasses classes;
    memset(&classes, 0, sizeof(classes));
    for (int i = 0; i < n; i++)
    {
        const x265gpu_pu_job& j = jobs[i];
        // the PU inside the picture, shapes the tiling handles, vectors that stay inside the padded plane
        if (j.w < 4 || j.h < 4 || j.w > 64 || j.h > 64 || (j.w & 3) || (j.h & 3) || j.x < 0 || j.y < 0 ||
            j.x + j.w > ctx->fullW || j.y + j.h > ctx->fullH || j.numCandidates < 0 || j.numCandidates > 12 ||
            j.mvminX > j.mvmaxX || j.mvminY > j.mvmaxY || (j.flags & 0xfe) || ((unsigned)j.flags >> 8) >= (unsigned)ctx->meCostTables ||
            j.x + j.mvminX < -ctx->fullMarginX + 8 || j.x + j.w + j.mvmaxX > ctx->fullW + ctx->fullMarginX - 16 ||
            j.y + j.mvminY < -ctx->fullMarginY + 8 || j.y + j.h + j.mvmaxY > ctx->fullH + ctx->fullMarginY - 8)
        {
            (void)hipStreamSynchronize(ctx->stream);      // the upload reads the caller's array
            return fail(ctx, X265GPU_ERR_ARG, "pu_motion_estimate: bad job");
        }
        const int segW = (j.w & 7) ? 4 : 8;
        const int tiles = (j.w / segW) * (j.h / 4);
        const int passes = (tiles + 15) / 16;
        passesOf[i] = passes;
        cls[i] = !pack ? 4 : tiles <= 2 ? 0 : tiles <= 4 ? 1 : tiles <= 8 ? 2 : tiles <= 32 ? 3 : 4;
        if (sea)
        {
            SeaShape sh;
            if (!sea_shape(j.w, j.h, sh))
            {
                (void)hipStreamSynchronize(ctx->stream);
                return fail(ctx, X265GPU_ERR_UNSUPPORTED, "pu_motion_estimate: no defined successive elimination for this PU shape");
            }
            // every block sum and every SAD the window can ask for (up to three columns past mvmaxX) lies inside
            // the padded picture
            int bw, bh, reachX = j.w + 16, reachY = j.h + 8;
            sea_plane_size(sh.plane, bw, bh);
            for (int k = 0; k < sh.terms; k++)
            {
                reachX = sh.sumX[k] + bw > reachX ? sh.sumX[k] + bw : reachX;
                reachY = sh.sumY[k] + bh > reachY ? sh.sumY[k] + bh : reachY;
            }
            if (j.x + j.mvmaxX + 3 + reachX > ctx->fullW + ctx->fullMarginX || j.y + j.mvmaxY + reachY > ctx->fullH + ctx->fullMarginY)
            {
                (void)hipStreamSynchronize(ctx->stream);
                return fail(ctx, X265GPU_ERR_ARG, "pu_motion_estimate: successive elimination window leaves the padded picture");
            }
            seaNeed |= 1u << sh.plane;
            cls[i] = tiles <= 32 ? 3 : 4;          // the elimination pass works with the whole wave
        }
        classes.count[cls[i]]++;
        if (cls[i] == 3)
            midPasses = passes > midPasses ? passes : midPasses;
        if (cls[i] == 4)
            maxPasses = passes > maxPasses ? passes : maxPasses;
    }
    classes.maxPasses[0] = midPasses;
    classes.maxPasses[1] = maxPasses;
    std::vector<int> order(n);
    {
        int at[5], sum = 0;
        for (int k = 0; k < 5; k++)
        {
            at[k] = sum;
            sum += classes.count[k];
        }
        const int bigAt = at[4];
        for (int i = 0; i < n; i++)
            order[at[cls[i]]++] = i;
        std::stable_sort(order.begin() + bigAt, order.end(), [&](int a, int b) { return passesOf[a] > passesOf[b]; });
    }
    maxPasses = maxPasses > midPasses ? maxPasses : midPasses;
    HIPCHK(hipMemcpyAsync(ctx->puBuf + jobBytes + outBytes, order.data(), orderBytes, hipMemcpyHostToDevice, ctx->stream));
    classes.order = (const int*)(ctx->puBuf + jobBytes + outBytes);
    PuFrame f = pu_frame(ctx, slotFenc, slotRef);
    f.method = searchMethod;
    f.merange = merange;
    f.subpelRefine = subpelRefine;
    f.maxPasses = maxPasses;
    if (sea)
    {
        const size_t planeElems = (size_t)ctx->fullStride * ctx->fullRows;
        if (!ctx->seaBuf)
            HIPCHK(hipMalloc((void**)&ctx->seaBuf, (SEA_PLANES + 1) * planeElems * sizeof(uint32_t)));
        if (ctx->seaSlot != slotRef || ctx->seaSerial != ctx->slots[slotRef].fullSerial)
        {
            ctx->seaSlot = slotRef;
            ctx->seaSerial = ctx->slots[slotRef].fullSerial;
            ctx->seaMask = 0;
        }
        uint32_t* planes[SEA_PLANES];
        const size_t originElems = (size_t)ctx->fullMarginY * ctx->fullStride + ctx->fullMarginX;
        for (int k = 0; k < SEA_PLANES; k++)
        {
            planes[k] = ctx->seaBuf + k * planeElems;
            f.seaSums[k] = planes[k] + originElems;
        }
        if (seaNeed & ~ctx->seaMask)
            HIPCHK(launch_sea_planes(ctx->stream, ctx->bpp, ctx->slots[slotRef].full, ctx->fullStride, ctx->fullRows,
                                     seaNeed & ~ctx->seaMask, ctx->seaBuf + SEA_PLANES * planeElems, planes));
        ctx->seaMask |= seaNeed;
    }
    HIPCHK(launch_pu_search(ctx->stream, ctx->searchAux, ctx->bpp, f, (const PuJob*)ctx->puBuf, classes, ctx->meCostCentre, ctx->meMaxMvd,
                            (PuResult*)(ctx->puBuf + jobBytes)));
    HIPCHK(hipMemcpyAsync(out, ctx->puBuf + jobBytes, sizeof(PuResult) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));    // also keeps `order` alive until its upload is done
    return X265GPU_OK;
}

/* ---- cuTree ---- */

int x265gpu_mcstf_motion_search(x265gpu_ctx* ctx, int32_t slotCur, int32_t numRefs, const int32_t* slotRefs, int32_t* mvs,
                                int32_t* error, int32_t* coarse)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || numRefs < 0 || numRefs > 16 || (numRefs && (!slotRefs || !mvs || !error)))
        return X265GPU_ERR_ARG;
    if (!numRefs)
        return X265GPU_OK;
    if (!ctx->cfg.bMcstfFilter || !ctx->fullStride)
        return fail(ctx, X265GPU_ERR_STATE, "mcstf_motion_search: context opened without bMcstfFilter and bKeepFullRes");
    if (!slot_ok(ctx, slotCur))
        return fail(ctx, X265GPU_ERR_ARG, "mcstf_motion_search: bad slot");
    for (int r = 0; r < numRefs; r++)
        if (!slot_ok(ctx, slotRefs[r]))
            return fail(ctx, X265GPU_ERR_ARG, "mcstf_motion_search: bad reference slot");
    const Geometry& g = ctx->g;
    const int W = g.srcWidth, H = g.srcHeight;
    // array sizes and strides of TemporalFilter::createRefPicInfo (temporalfilter.cpp:397-409)
    const size_t n16 = (size_t)(W / 16) * (H / 16), n4 = (size_t)(W / 4) * (H / 4);
    if (!n16)
        return fail(ctx, X265GPU_ERR_UNSUPPORTED, "mcstf_motion_search: picture smaller than one block");
    const size_t perRef = align256(3 * n16 * 2 * sizeof(int32_t)) + align256(n4 * 2 * sizeof(int32_t)) +
                          2 * align256(n4 * sizeof(int32_t)) + align256(n4 * 4 * sizeof(int32_t));
    const size_t total = perRef * numRefs;
    if (total > ctx->mcstfBufSize)
    {
        if (ctx->mcstfBuf)
        {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipFree(ctx->mcstfBuf));
            ctx->mcstfBuf = nullptr;
            ctx->mcstfBufSize = 0;
        }
        HIPCHK(hipMalloc((void**)&ctx->mcstfBuf, total));
        ctx->mcstfBufSize = total;
    }
    HIPCHK(hipMemsetAsync(ctx->mcstfBuf, 0, total, ctx->stream));
    int rc = ensure_job_buf(ctx, 4 * align256(sizeof(McstfRef) * numRefs));
    if (rc)
        return rc;

    struct RefArrays { int32_t* coarse[3]; int32_t* mvs; int32_t* error; int32_t* least; int32_t* spec; };
    std::vector<RefArrays> arr(numRefs);
    for (int r = 0; r < numRefs; r++)
    {
        char* p = ctx->mcstfBuf + perRef * r;
        for (int k = 0; k < 3; k++)
            arr[r].coarse[k] = (int32_t*)p + k * n16 * 2;
        p += align256(3 * n16 * 2 * sizeof(int32_t));
        arr[r].mvs = (int32_t*)p;
        p += align256(n4 * 2 * sizeof(int32_t));
        arr[r].error = (int32_t*)p;
        p += align256(n4 * sizeof(int32_t));
        arr[r].least = (int32_t*)p;
        p += align256(n4 * sizeof(int32_t));
        arr[r].spec = (int32_t*)p;
    }
    const size_t fullOrigin = ((size_t)ctx->fullMarginY * ctx->fullStride + ctx->fullMarginX) * ctx->bpp;
    const Slot& sc = ctx->slots[slotCur];
    const int rowSize = 16;                                        // PARALLEL_ME_ROWSIZE, slicetype.h:67
    const int levelHeight[4] = { H >> 2, H >> 1, H, H };          // slicetype.cpp:2282
    for (int level = 0; level < 4; level++)
    {
        McstfLevel lv;
        memset(&lv, 0, sizeof(lv));
        lv.depth = ctx->cfg.internalBitDepth;
        const int numBlockRows = (levelHeight[level] + rowSize - 1) / rowSize;
        // the arguments of slicetype.cpp:4227-4233
        if (level == 0)
        {
            lv.src = plane4_origin(ctx, sc.planes4);
            lv.stride = g.stride / 2;
            lv.width = g.width / 2;
            lv.height = g.lines / 2;
        }
        else if (level == 1)
        {
            lv.src = plane_origin(ctx, sc.planes);
            lv.stride = g.stride;
            lv.width = g.width;
            lv.height = g.lines;
        }
        else
        {
            lv.src = sc.full + fullOrigin;
            lv.stride = ctx->fullStride;
            lv.width = W;
            lv.height = H;
        }
        const int bs = level == 3 ? 8 : 16;
        lv.blocksX = lv.width / bs;
        // a row task starts every 16 lines and holds the blocks that end inside it and inside the picture
        // (temporalfilter.cpp:653-661, 762-771)
        const int rowsInPicture = lv.height / bs, rowsInTasks = numBlockRows * (rowSize / bs);
        lv.blocksY = rowsInPicture < rowsInTasks ? rowsInPicture : rowsInTasks;
        lv.mvStride = level == 3 ? W / 4 : W / 16;
        lv.prevStride = W / 16;
        lv.factor = level == 3 ? 1 : 2;
        std::vector<McstfRef> refs(numRefs);
        for (int r = 0; r < numRefs; r++)
        {
            const Slot& sr = ctx->slots[slotRefs[r]];
            refs[r].buf = level == 0 ? plane4_origin(ctx, sr.planes4) : level == 1 ? plane_origin(ctx, sr.planes)
                                                                                   : sr.full + fullOrigin;
            refs[r].mvs = level == 3 ? arr[r].mvs : arr[r].coarse[level];
            refs[r].prev = level == 0 ? nullptr : arr[r].coarse[level - 1];
            refs[r].least = arr[r].least;
            refs[r].error = arr[r].error;
            refs[r].spec = arr[r].spec;
        }
        const size_t off = level * align256(sizeof(McstfRef) * numRefs);
        rc = upload_jobs(ctx, off, refs.data(), sizeof(McstfRef) * numRefs, level == 0);
        if (rc)
            return rc;
        HIPCHK(launch_mcstf_level(ctx->stream, ctx->bpp, level, lv, (const McstfRef*)(ctx->jobBuf + off), numRefs));
    }
    for (int r = 0; r < numRefs; r++)
    {
        HIPCHK(hipMemcpyAsync(mvs + (size_t)r * n4 * 2, arr[r].mvs, n4 * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(error + (size_t)r * n4, arr[r].error, n4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (coarse)
            HIPCHK(hipMemcpyAsync(coarse + (size_t)r * 3 * n16 * 2, arr[r].coarse[0], 3 * n16 * 2 * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_cutree_clear(x265gpu_ctx* ctx, int32_t slot)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slot))
        return X265GPU_ERR_ARG;
    x265gpu_ctx::CuTreeOp op;
    memset(&op, 0, sizeof(op));
    op.clear = true;
    op.cells = ctx->slots[slot].propagate;
    op.count = (int)ctx->g.cuCount;
    ctx->cutreeOps.push_back(op);
    return X265GPU_OK;
}

int x265gpu_cutree_swap(x265gpu_ctx* ctx, int32_t slotA, int32_t slotB)
{
    if (!ctx || !slot_ok(ctx, slotA) || !slot_ok(ctx, slotB))
        return X265GPU_ERR_ARG;
    DeviceScope deviceScope(ctx);
    int rc = cutree_flush(ctx);            // queued passes hold the arrays' addresses
    if (rc)
        return rc;
    std::swap(ctx->slots[slotA].propagate, ctx->slots[slotB].propagate);
    return X265GPU_OK;
}

int x265gpu_cutree_propagate(x265gpu_ctx* ctx, int32_t slotP0, int32_t slotP1, int32_t slotB, int32_t d0, int32_t d1,
                             int32_t referenced, double fpsFactor, int32_t bipredWeight)
{
    DeviceScope deviceScope(ctx);
    if (ctx)
    {
        const int rcAq = aq_finish_pending(ctx);      // the scale factors this pass reads
        if (rcAq)
            return rcAq;
    }
    if (!ctx || !slot_ok(ctx, slotP0) || !slot_ok(ctx, slotP1) || !slot_ok(ctx, slotB) || d0 < 1 || d0 >= ctx->D ||
        d1 < 0 || d1 >= ctx->D || bipredWeight < 0 || bipredWeight > 64)
        return X265GPU_ERR_ARG;
    Slot& sb = ctx->slots[slotB];
    if (!sb.hasAq)
        return fail(ctx, X265GPU_ERR_STATE, "cutree_propagate: the frame has no invQscaleFactor (x265gpu_lowres_set_aq)");
    const size_t nb = ctx->g.cuCount;
    PropagateJob job;
    memset(&job, 0, sizeof(job));
    job.intraCost = sb.intraCost;
    job.lowresCosts = sb.lowresCosts + ((size_t)d0 * ctx->D + d1) * nb;
    job.invQscale = sb.invQscale;
    job.propagateIn = referenced ? sb.propagate : nullptr;
    job.mvs[0] = sb.mvs + ((size_t)0 * ctx->D + d0) * nb * 2;
    job.mvs[1] = sb.mvs + ((size_t)1 * ctx->D + d1) * nb * 2;
    job.refCosts[0] = ctx->slots[slotP0].propagate;
    job.refCosts[1] = ctx->slots[slotP1].propagate;
    job.W = ctx->g.W;
    job.H = ctx->g.H;
    job.bipredWeight = bipredWeight;
    job.fps = fpsFactor / 256;
    // an unreferenced frame's own cells are a zeroed row in the reference (slicetype.cpp:3784-3785):
    // nothing reads them again before the next clear, but keep the visible state the same
    x265gpu_ctx::CuTreeOp op;
    memset(&op, 0, sizeof(op));
    if (!referenced)
    {
        op.clear = true;
        op.cells = sb.propagate;
        op.count = ctx->g.W;
        ctx->cutreeOps.push_back(op);
    }
    op.clear = false;
    op.cells = nullptr;
    op.count = 0;
    op.job = job;
    ctx->cutreeOps.push_back(op);
    return X265GPU_OK;
}

// Launches the queued cuTree passes.  A pass reads its own frame's totals and adds to its references' totals; adds to
// one array commute, so passes only have to wait for the clears of the arrays they touch, for every add into the
// array they read, and (a clear) for everything that touched the array before.  Each pass gets the lowest level its
// predecessors allow and a level is one launch: the unreferenced B frames of a whole walk go together, and what
// remains is the chain from each P frame to the one before it.
static int cutree_flush(x265gpu_ctx* ctx)
{
    std::vector<x265gpu_ctx::CuTreeOp>& ops = ctx->cutreeOps;
    if (ops.empty())
        return X265GPU_OK;
    const size_t n = ops.size();
    std::vector<int> level(n, 0);
    int levels = 0;
    for (size_t j = 0; j < n; j++)
    {
        const x265gpu_ctx::CuTreeOp& b = ops[j];
        int lv = 0;
        for (size_t i = 0; i < j; i++)
        {
            const x265gpu_ctx::CuTreeOp& a = ops[i];
            bool after;
            if (a.clear && b.clear)
                after = a.cells == b.cells;
            else if (a.clear)
                after = a.cells == b.job.propagateIn || a.cells == b.job.refCosts[0] || a.cells == b.job.refCosts[1];
            else if (b.clear)
                after = b.cells == a.job.propagateIn || b.cells == a.job.refCosts[0] || b.cells == a.job.refCosts[1];
            else
                after = (b.job.propagateIn && (b.job.propagateIn == a.job.refCosts[0] || b.job.propagateIn == a.job.refCosts[1])) ||
                        (a.job.propagateIn && (a.job.propagateIn == b.job.refCosts[0] || a.job.propagateIn == b.job.refCosts[1]));
            if (after && level[i] + 1 > lv)
                lv = level[i] + 1;
        }
        level[j] = lv;
        levels = lv + 1 > levels ? lv + 1 : levels;
    }
    // device copies, level after level: the passes' jobs, then the clears' arrays and lengths
    std::vector<PropagateJob> jobs;
    std::vector<unsigned long long*> cells;
    std::vector<int> counts;
    std::vector<int> jobAt(levels + 1, 0), clearAt(levels + 1, 0);
    for (int lv = 0; lv < levels; lv++)
    {
        jobAt[lv] = (int)jobs.size();
        clearAt[lv] = (int)cells.size();
        for (size_t i = 0; i < n; i++)
            if (level[i] == lv)
            {
                if (ops[i].clear)
                {
                    cells.push_back(ops[i].cells);
                    counts.push_back(ops[i].count);
                }
                else
                    jobs.push_back(ops[i].job);
            }
    }
    jobAt[levels] = (int)jobs.size();
    clearAt[levels] = (int)cells.size();
    const size_t jobBytes = align256(sizeof(PropagateJob) * jobs.size()), cellBytes = align256(sizeof(void*) * cells.size());
    const size_t total = jobBytes + cellBytes + align256(sizeof(int) * counts.size());
    if (total > ctx->cutreeBufBytes)
    {
        if (ctx->cutreeBuf)
        {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipFree(ctx->cutreeBuf));
            ctx->cutreeBuf = nullptr;
            ctx->cutreeBufBytes = 0;
        }
        HIPCHK(hipMalloc((void**)&ctx->cutreeBuf, 2 * total));
        ctx->cutreeBufBytes = 2 * total;
    }
    ops.clear();
    if (!jobs.empty())
        HIPCHK(hipMemcpyAsync(ctx->cutreeBuf, jobs.data(), sizeof(PropagateJob) * jobs.size(), hipMemcpyHostToDevice, ctx->stream));
    if (!cells.empty())
    {
        HIPCHK(hipMemcpyAsync(ctx->cutreeBuf + jobBytes, cells.data(), sizeof(void*) * cells.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->cutreeBuf + jobBytes + cellBytes, counts.data(), sizeof(int) * counts.size(), hipMemcpyHostToDevice,
                              ctx->stream));
    }
    const PropagateJob* dJobs = (const PropagateJob*)ctx->cutreeBuf;
    unsigned long long* const* dCells = (unsigned long long* const*)(ctx->cutreeBuf + jobBytes);
    const int* dCounts = (const int*)(ctx->cutreeBuf + jobBytes + cellBytes);
    for (int lv = 0; lv < levels; lv++)
    {
        const int nc = clearAt[lv + 1] - clearAt[lv], nj = jobAt[lv + 1] - jobAt[lv];
        if (nc)
        {
            int maxCount = 0;
            for (int k = clearAt[lv]; k < clearAt[lv + 1]; k++)
                maxCount = counts[k] > maxCount ? counts[k] : maxCount;
            HIPCHK(launch_cutree_clear_batch(ctx->stream, dCells + clearAt[lv], dCounts + clearAt[lv], nc, maxCount));
        }
        if (nj)
            HIPCHK(launch_cutree_propagate_batch(ctx->stream, dJobs + jobAt[lv], nj, ctx->g.W, ctx->g.H));
    }
    // the uploads above read these vectors
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

namespace {
int fetch_propagate(x265gpu_ctx* ctx, int slot, std::vector<unsigned long long>& acc)
{
    acc.resize(ctx->g.cuCount);
    HIPCHK(hipMemcpyAsync(acc.data(), ctx->slots[slot].propagate, sizeof(unsigned long long) * acc.size(),
                          hipMemcpyDeviceToHost, ctx->stream));
    return X265GPU_OK;
}
} // namespace

int x265gpu_cutree_get_propagate_cost(x265gpu_ctx* ctx, int32_t slot, uint16_t* propagateCost)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slot) || !propagateCost)
        return X265GPU_ERR_ARG;
    FINISH_AQ(ctx);
    std::vector<unsigned long long> acc;
    int rc = fetch_propagate(ctx, slot, acc);
    if (rc)
        return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < acc.size(); i++)
        propagateCost[i] = (uint16_t)(acc[i] < 65535 ? acc[i] : 65535);
    return X265GPU_OK;
}

namespace {

// What cuTreeFinish reads of a frame: propagated totals, intra costs and scale factors of every block, brought to
// the context's page-locked landing area in three transfers and one wait.
int fetch_finish_inputs(x265gpu_ctx* ctx, int32_t slot, const unsigned long long** acc, const int32_t** intra, const int32_t** invQ)
{
    Slot& s = ctx->slots[slot];
    const size_t nb = ctx->g.cuCount;
    const size_t accBytes = align256(sizeof(unsigned long long) * nb), arrBytes = align256(sizeof(int32_t) * nb);
    const size_t total = accBytes + 2 * arrBytes;
    if (total > ctx->finishHostBytes)
    {
        if (ctx->finishHost)
            HIPCHK(hipHostFree(ctx->finishHost));
        ctx->finishHost = nullptr;
        ctx->finishHostBytes = 0;
        HIPCHK(hipHostMalloc((void**)&ctx->finishHost, total, hipHostMallocDefault));
        ctx->finishHostBytes = total;
    }
    char* h = ctx->finishHost;
    HIPCHK(hipMemcpyAsync(h, s.propagate, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(h + accBytes, s.intraCost, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(h + accBytes + arrBytes, s.invQscale, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *acc = (const unsigned long long*)h;
    *intra = (const int32_t*)(h + accBytes);
    *invQ = (const int32_t*)(h + accBytes + arrBytes);
    return X265GPU_OK;
}

int cutree_finish(x265gpu_ctx* ctx, int32_t slot, bool qg8, int32_t fpsFactor, double weightDelta, double cuTreeStrength,
                  const double* qpAqOffset, double* qpCuTreeOffset)
{
    if (!ctx || !slot_ok(ctx, slot) || !qpAqOffset || !qpCuTreeOffset)
        return X265GPU_ERR_ARG;
    Slot& s = ctx->slots[slot];
    if (!s.hasAq)
        return fail(ctx, X265GPU_ERR_STATE, "cutree_finish: the frame has no invQscaleFactor (x265gpu_lowres_set_aq)");
    const size_t nb = ctx->g.cuCount;
    const unsigned long long* acc;
    const int32_t *intra, *invQ;
    int rc = fetch_finish_inputs(ctx, slot, &acc, &intra, &invQ);
    if (rc)
        return rc;
    if (qg8)
        cutree_finish_offsets_qg8(ctx->g.W, ctx->g.H, intra, invQ, acc, fpsFactor, weightDelta, cuTreeStrength, qpAqOffset,
                                  qpCuTreeOffset);
    else
        cutree_finish_offsets((int)nb, intra, invQ, acc, fpsFactor, weightDelta, cuTreeStrength, qpAqOffset, qpCuTreeOffset);
    return X265GPU_OK;
}

} // namespace

int x265gpu_cutree_finish(x265gpu_ctx* ctx, int32_t slot, int32_t fpsFactor, double weightDelta, double cuTreeStrength,
                          const double* qpAqOffset, double* qpCuTreeOffset)
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);
    return cutree_finish(ctx, slot, false, fpsFactor, weightDelta, cuTreeStrength, qpAqOffset, qpCuTreeOffset);
}

int x265gpu_cutree_finish_qg8(x265gpu_ctx* ctx, int32_t slot, int32_t fpsFactor, double weightDelta, double cuTreeStrength,
                              const double* qpAqOffset, double* qpCuTreeOffset)
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);
    return cutree_finish(ctx, slot, true, fpsFactor, weightDelta, cuTreeStrength, qpAqOffset, qpCuTreeOffset);
}

int x265gpu_cutree_finish_hevc(x265gpu_ctx* ctx, int32_t slot, int32_t fpsFactor, double weightDelta, double cuTreeStrength,
                               int32_t qgSize, int32_t numLayers, x265gpu_hevc_aq_layer* layers, double* dCuTreeOffset8x8)
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);
    if (!ctx || !slot_ok(ctx, slot) || numLayers < 1 || numLayers > 4 || !layers)
        return X265GPU_ERR_ARG;
    const bool qg8 = qgSize == 8;
    if (qg8 && !dCuTreeOffset8x8)
        return fail(ctx, X265GPU_ERR_ARG, "cutree_finish_hevc: qgSize 8 needs the dCuTreeOffset8x8 array");
    Slot& s = ctx->slots[slot];
    if (!s.hasAq)
        return fail(ctx, X265GPU_ERR_STATE, "cutree_finish_hevc: the frame has no invQscaleFactor (x265gpu_lowres_set_aq)");
    int partW[4], partH[4], numW[4], numH[4];
    const double* qp[4];
    double* out[4];
    for (int d = 0; d < numLayers; d++)
    {
        const x265gpu_hevc_aq_layer& l = layers[d];
        if (l.aqPartWidth < (qg8 ? 8 : 16) || l.aqPartHeight < (qg8 ? 8 : 16) || l.numAQPartInWidth < 1 || l.numAQPartInHeight < 1 ||
            !l.dQpOffset || !l.dCuTreeOffset)
            // (a part may lie wholly outside the picture: the doubled lowres grid of the finest layer has such rows
            // and columns whenever the picture is not a multiple of 16, and the reference gives them 0 / 0)
            return fail(ctx, X265GPU_ERR_ARG, "cutree_finish_hevc: bad layer");
        partW[d] = l.aqPartWidth;
        partH[d] = l.aqPartHeight;
        numW[d] = l.numAQPartInWidth;
        numH[d] = l.numAQPartInHeight;
        qp[d] = l.dQpOffset;
        out[d] = l.dCuTreeOffset;
    }
    const unsigned long long* acc;
    const int32_t *intra, *invQ;
    int rc = fetch_finish_inputs(ctx, slot, &acc, &intra, &invQ);
    if (rc)
        return rc;
    cutree_finish_hevc(ctx->g.W, ctx->g.H, ctx->cfg.sourceWidth, ctx->cfg.sourceHeight, qg8, intra, invQ, acc,
                       fpsFactor, weightDelta, cuTreeStrength, numLayers, partW, partH, numW, numH, qp, out, dCuTreeOffset8x8);
    return X265GPU_OK;
}

namespace {

// A caller's 4:2:0 source picture as the kernels read it: device planes are used in place,
// host planes are copied tight into the context's staging buffer first.
struct YuvOnDevice
{
    const void* y; const void* u; const void* v;
    long long strideY, strideC;
};

int stage_yuv(x265gpu_ctx* ctx, const void* y, intptr_t strideY, const void* u, const void* v, intptr_t strideC,
              int32_t memKind, YuvOnDevice& out)
{
    const int w = ctx->cfg.sourceWidth, h = ctx->cfg.sourceHeight, cw = w >> 1, ch = h >> 1;
    if (memKind != X265GPU_MEM_HOST)
    {
        out = { y, u, v, (long long)strideY, (long long)strideC };
        return X265GPU_OK;
    }
    const size_t lumaBytes = (size_t)w * h * ctx->bpp, chromaBytes = (size_t)cw * ch * ctx->bpp;
    int rc = ensure_src_stage(ctx, align256(lumaBytes) + 2 * align256(chromaBytes));
    if (rc)
        return rc;
    char* dy = ctx->srcStage;
    char* du = dy + align256(lumaBytes);
    char* dv = du + align256(chromaBytes);
    HIPCHK(hipMemcpy2DAsync(dy, (size_t)w * ctx->bpp, y, (size_t)strideY * ctx->bpp, (size_t)w * ctx->bpp, h,
                            hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpy2DAsync(du, (size_t)cw * ctx->bpp, u, (size_t)strideC * ctx->bpp, (size_t)cw * ctx->bpp, ch,
                            hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpy2DAsync(dv, (size_t)cw * ctx->bpp, v, (size_t)strideC * ctx->bpp, (size_t)cw * ctx->bpp, ch,
                            hipMemcpyHostToDevice, ctx->stream));
    out = { dy, du, dv, (long long)w, (long long)cw };
    return X265GPU_OK;
}

// the six frame sums out of the energy kernel's sharded accumulators
void sum_wp_shards(const unsigned long long* shards, unsigned long long wp[6])
{
    for (int i = 0; i < 6; i++)
        wp[i] = 0;
    for (int s = 0; s < AQ_WP_SHARDS; s++)
        for (int i = 0; i < 6; i++)
            wp[i] += shards[s * AQ_WP_SHARD_STRIDE + i];
}

// One device pass over a staged 4:2:0 picture: block energies (blockSize 16 or 8), optional block luma
// sums and the six frame accumulators, copied to the host.  The block loop runs while blockX < visitW and
// blockY < visitH; arrays hold ceil(visitW / blockSize) * ceil(visitH / blockSize) entries in loop order.
int aq_energy_pass(x265gpu_ctx* ctx, const YuvOnDevice& pic, int blockSize, int visitW, int visitH, uint32_t* energy,
                   uint32_t* lumaSum, unsigned long long wp[6])
{
    const int blocks = ((visitW + blockSize - 1) / blockSize) * ((visitH + blockSize - 1) / blockSize);
    for (int i = 0; i < 6; i++)
        wp[i] = 0;
    if (blocks < 1)
        return X265GPU_OK;
    const size_t arrayBytes = align256(sizeof(uint32_t) * (size_t)blocks);
    int rc = ensure_pu_buf(ctx, 2 * arrayBytes + AQ_WP_BYTES);
    if (rc)
        return rc;
    AqJob job;
    memset(&job, 0, sizeof(job));
    job.width = ctx->cfg.sourceWidth;
    job.height = ctx->cfg.sourceHeight;
    job.visitWidth = visitW;
    job.visitHeight = visitH;
    job.blockSize = blockSize;
    job.luma = pic.y;
    job.u = pic.u;
    job.v = pic.v;
    job.lumaStride = pic.strideY;
    job.chromaStride = pic.strideC;
    job.energy = (uint32_t*)ctx->puBuf;
    job.lumaSum = lumaSum ? (uint32_t*)(ctx->puBuf + arrayBytes) : nullptr;
    job.wp = (unsigned long long*)(ctx->puBuf + 2 * arrayBytes);
    HIPCHK(hipMemsetAsync(job.wp, 0, AQ_WP_BYTES, ctx->stream));
    HIPCHK(launch_aq_energy(ctx->stream, ctx->bpp, job));
    HIPCHK(hipMemcpyAsync(energy, job.energy, sizeof(uint32_t) * (size_t)blocks, hipMemcpyDeviceToHost, ctx->stream));
    if (lumaSum)
        HIPCHK(hipMemcpyAsync(lumaSum, job.lumaSum, sizeof(uint32_t) * (size_t)blocks, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long shards[AQ_WP_SHARDS * AQ_WP_SHARD_STRIDE];
    HIPCHK(hipMemcpyAsync(shards, job.wp, AQ_WP_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    sum_wp_shards(shards, wp);
    return X265GPU_OK;
}

int stage_aq_picture(x265gpu_ctx* ctx, const char* who, const void* y, intptr_t strideY, const void* u, const void* v,
                     intptr_t strideC, int32_t memKind, YuvOnDevice& pic)
{
    const int w = ctx->cfg.sourceWidth, h = ctx->cfg.sourceHeight, cw = w >> 1, ch = h >> 1;
    if (strideY < w || strideC < cw || cw < 1 || ch < 1)
        return fail(ctx, X265GPU_ERR_ARG, (std::string(who) + ": bad stride or picture size").c_str());
    return stage_yuv(ctx, y, strideY, u, v, strideC, memKind, pic);
}

} // namespace

int x265gpu_aq_energy(x265gpu_ctx* ctx, const void* y, intptr_t strideY, const void* u, const void* v, intptr_t strideC,
                      int32_t memKind, uint32_t* energy, uint64_t wpSum[3], uint64_t wpSsd[3])
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !y || !u || !v || !energy || !wpSum || !wpSsd)
        return X265GPU_ERR_ARG;
    if (ctx->oddSize)
        return fail(ctx, X265GPU_ERR_UNSUPPORTED, "aq_energy: 4:2:0 pictures have even sizes (the context was opened with an odd one)");
    unsigned long long wp[6];
    YuvOnDevice pic;
    int rc = stage_aq_picture(ctx, "aq_energy", y, strideY, u, v, strideC, memKind, pic);
    if (rc)
        return rc;
    rc = aq_energy_pass(ctx, pic, 16, ctx->cfg.sourceWidth, ctx->cfg.sourceHeight, energy, nullptr, wp);
    if (rc)
        return rc;
    for (int i = 0; i < 3; i++)
    {
        wpSum[i] = wp[i];
        wpSsd[i] = wp[3 + i];
    }
    return X265GPU_OK;
}

namespace {
// The edge mode's picture pass (edgeFilter + edgeDensityCu, source/encoder/slicetype.cpp:191-263) over a staged
// picture, in bands of tile rows so that the list of pixels left to the host's atan2f always fits its buffer.
// edgeDensityCu runs the edge picture's blocks through acEnergyVar, which adds their sums to plane 0 of wp_sum /
// wp_ssd: wp[0] and wp[3] receive those as well.
int aq_edge_pass(x265gpu_ctx* ctx, const YuvOnDevice& pic, int B, int w, int h, int cols, int blocks,
                 std::vector<uint32_t>& edgeDensity, std::vector<uint32_t>& avgAngle, unsigned long long wp[6])
{
    const int tileRowsAll = (h + 15) / 16, tileCols = (w + 15) / 16;
    const size_t listCap = (size_t)1 << 21;                                  // records per band
    int bandRows = (int)(listCap / ((size_t)tileCols * 256));
    if (const char* e = getenv("X265GPU_AQ_EDGE_BAND_ROWS"))      // smaller bands (tests)
        bandRows = atoi(e) < bandRows ? atoi(e) : bandRows;
    bandRows = bandRows < 1 ? 1 : bandRows;
    const size_t arrayBytes = align256(sizeof(uint32_t) * (size_t)blocks);
    const size_t bandPixels = (size_t)bandRows * tileCols * 256;
    const size_t total = 3 * arrayBytes + 256 + bandPixels * 3 * sizeof(uint32_t);
    int rc = ensure_pu_buf(ctx, total);
    if (rc)
        return rc;
    AqEdgeJob job;
    memset(&job, 0, sizeof(job));
    job.luma = pic.y;
    job.lumaStride = pic.strideY;
    job.width = w;
    job.height = h;
    job.blockSize = B;
    job.blocksX = cols;
    job.theta180 = edge_theta(0, -1);
    job.theta90pos = edge_theta(1, 0);
    job.theta90neg = edge_theta(-1, 0);
    job.edgeSum = (uint32_t*)ctx->puBuf;
    job.edgeSsd = (uint32_t*)(ctx->puBuf + arrayBytes);
    job.thetaSum = (uint32_t*)(ctx->puBuf + 2 * arrayBytes);
    job.listCount = (uint32_t*)(ctx->puBuf + 3 * arrayBytes);
    job.list = (uint32_t*)(ctx->puBuf + 3 * arrayBytes + 256);
    std::vector<uint32_t> extra((size_t)blocks, 0), records;
    for (int row0 = 0; row0 < tileRowsAll; row0 += bandRows)
    {
        const int rows = bandRows < tileRowsAll - row0 ? bandRows : tileRowsAll - row0;
        job.tileRow0 = row0;
        HIPCHK(hipMemsetAsync(job.listCount, 0, sizeof(uint32_t), ctx->stream));
        HIPCHK(launch_aq_edge(ctx->stream, ctx->bpp, ctx->cfg.internalBitDepth, job, rows));
        uint32_t count = 0;
        HIPCHK(hipMemcpyAsync(&count, job.listCount, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (!count)
            continue;
        records.resize(3 * (size_t)count);
        HIPCHK(hipMemcpyAsync(records.data(), job.list, 3 * (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < count; i++)
            extra[records[3 * (size_t)i]] += (uint32_t)edge_theta((int)records[3 * (size_t)i + 1], (int)records[3 * (size_t)i + 2]);
    }
    std::vector<uint32_t> sums(3 * (size_t)blocks);
    for (int k = 0; k < 3; k++)
        HIPCHK(hipMemcpyAsync(sums.data() + (size_t)k * blocks, ctx->puBuf + k * arrayBytes, sizeof(uint32_t) * (size_t)blocks,
                              hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    edgeDensity.resize((size_t)blocks);
    avgAngle.resize((size_t)blocks);
    const int shift = B == 8 ? 6 : 8;
    for (int i = 0; i < blocks; i++)
    {
        const uint32_t sum = sums[i], ssd = sums[(size_t)blocks + i];
        wp[0] += sum;
        wp[3] += ssd;
        edgeDensity[i] = ssd - (uint32_t)(((unsigned long long)sum * sum) >> shift);      // acEnergyVar, :51-59
        avgAngle[i] = (sums[2 * (size_t)blocks + i] + extra[i]) / (uint32_t)(B * B);        // findAvgAngle, :228-240
    }
    return X265GPU_OK;
}
} // namespace

int x265gpu_aq_frame(x265gpu_ctx* ctx, int32_t slot, const void* y, intptr_t strideY, const void* u, const void* v,
                     intptr_t strideC, int32_t memKind, const x265gpu_aq_params* params, const float* quantOffsets,
                     double* qpAqOffset, double* qpCuTreeOffset, int32_t* invQscaleFactor, uint64_t wpSum[3],
                     uint64_t wpSsd[3])
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);
    if (!ctx || !params || !y || !u || !v || !wpSum || !wpSsd)
        return X265GPU_ERR_ARG;
    if (ctx->oddSize)
        return fail(ctx, X265GPU_ERR_UNSUPPORTED, "aq_frame: 4:2:0 pictures have even sizes (the context was opened with an odd one)");
    if (params->aqMode < 0 || params->aqMode > 4)
        return fail(ctx, X265GPU_ERR_ARG, "aq_frame: aqMode must be 0..4");
    if (params->qgSize != 0 && params->qgSize != 8 && params->qgSize != 16 && params->qgSize != 32 && params->qgSize != 64)
        return fail(ctx, X265GPU_ERR_ARG, "aq_frame: qgSize must be 8, 16, 32 or 64 (0 = 32)");
    if (slot >= 0 && !slot_ok(ctx, slot))
        return fail(ctx, X265GPU_ERR_ARG, "aq_frame: bad slot");
    const int w = ctx->cfg.sourceWidth, h = ctx->cfg.sourceHeight;
    const bool qg8 = params->qgSize == 8;
    const int B = qg8 ? 8 : 16;                                   // loopIncr (slicetype.cpp:461-474)
    const int cols = (w + B - 1) / B, rows = (h + B - 1) / B, visited = cols * rows;
    // blockCount (:463, :470): with qgSize 8 the full-resolution grid is twice the lowres grid each way, which
    // can be one column / row more than the block loop visits; averages divide by it all the same
    const int fullCols = 2 * ctx->g.W, fullRows = 2 * ctx->g.H;
    const int blockCount = qg8 ? fullCols * fullRows : visited;
    const bool offsets = !params->bSkipOffsets && params->aqMode != 0;
    const bool mapped = offsets && params->aqStrength != 0;     // slicetype.cpp:486: the energy-driven modes
    const bool hevcAq = mapped && params->bHevcAq;              // :517: only inside the energy-driven branch
    if (offsets && (!invQscaleFactor || (!hevcAq && (!qpAqOffset || !qpCuTreeOffset))))
        return X265GPU_ERR_ARG;
    if (hevcAq)
    {
        if (params->numLayers < 1 || params->numLayers > 4 || !params->layers)
            return fail(ctx, X265GPU_ERR_ARG, "aq_frame: bHevcAq needs 1..4 layers");
        for (int i = 0; i < params->numLayers; i++)
        {
            const x265gpu_hevc_aq_layer& l = params->layers[i];
            if (l.aqPartWidth < 2 || l.aqPartHeight < 2 || l.aqPartWidth > 64 || l.aqPartHeight > 64 || !l.dActivity ||
                !l.dQpOffset || !l.dCuTreeOffset ||
                (long long)l.numAQPartInWidth * l.numAQPartInHeight <
                    (long long)((w + l.aqPartWidth - 1) / l.aqPartWidth) * ((h + l.aqPartHeight - 1) / l.aqPartHeight))
                return fail(ctx, X265GPU_ERR_ARG, "aq_frame: bad HEVC AQ layer");
        }
        const x265gpu_hevc_aq_layer& last = params->layers[params->numLayers - 1];
        if (last.aqPartWidth != B || last.aqPartHeight != B)
            return fail(ctx, X265GPU_ERR_ARG, "aq_frame: the last HEVC AQ layer must have parts of the AQ block size");
    }
    if (qg8 && offsets && !params->invQscaleFactor8x8)
        return fail(ctx, X265GPU_ERR_ARG, "aq_frame: qgSize 8 needs the invQscaleFactor8x8 output");
    if (params->bBlockVariance && (!params->blockVariance || !params->frameVariance))
        return fail(ctx, X265GPU_ERR_ARG, "aq_frame: bBlockVariance needs the blockVariance and frameVariance outputs");
    if (slot >= 0 && !qg8 && visited != ctx->g.cuCount)
        return fail(ctx, X265GPU_ERR_ARG, "aq_frame: 16x16 block grid differs from the slot's 8x8 lowres grid");

    unsigned long long wp[6] = { 0, 0, 0, 0, 0, 0 };
    YuvOnDevice pic;
    bool staged = false;
    std::vector<uint32_t> energy;
    // acEnergyCu runs once per block when a mode maps energies (:547, :609) or weighted prediction wants the
    // frame sums (:508-513, :665-670); otherwise not at all
    if (mapped || params->bWeightedPred)
    {
        energy.resize((size_t)visited);
        std::vector<uint32_t> lumaSum(params->bHDR10Opt && mapped ? (size_t)visited : 0);
        int rc = stage_aq_picture(ctx, "aq_frame", y, strideY, u, v, strideC, memKind, pic);
        if (rc)
            return rc;
        staged = true;
        rc = aq_energy_pass(ctx, pic, B, w, h, energy.data(), lumaSum.empty() ? nullptr : lumaSum.data(), wp);
        if (rc)
            return rc;
        if (hevcAq)
        {
            for (int i = 0; i < params->numLayers; i++)
            {
                x265gpu_hevc_aq_layer& l = params->layers[i];
                const size_t parts = (size_t)((w + l.aqPartWidth - 1) / l.aqPartWidth) * ((h + l.aqPartHeight - 1) / l.aqPartHeight);
                rc = ensure_pu_buf(ctx, parts * 8 * sizeof(unsigned long long));
                if (rc)
                    return rc;
                std::vector<unsigned long long> quadrants(parts * 8);
                HIPCHK(launch_aq_quadrants(ctx->stream, ctx->bpp, pic.y, pic.strideY, w, h, l.aqPartWidth, l.aqPartHeight,
                                           (unsigned long long*)ctx->puBuf));
                HIPCHK(hipMemcpyAsync(quadrants.data(), ctx->puBuf, parts * 8 * sizeof(unsigned long long),
                                      hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                l.dAvgActivity = hevc_aq_layer(w, h, l.aqPartWidth, l.aqPartHeight, l.numAQPartInWidth, l.numAQPartInHeight,
                                               params->qpAdaptationRange, quadrants.data(), l.dActivity, l.dQpOffset,
                                               l.dCuTreeOffset);
            }
            const x265gpu_hevc_aq_layer& last = params->layers[params->numLayers - 1];
            for (int i = 0; i < visited; i++)                     // :441-447
                invQscaleFactor[i] = aq_exp2fix8(last.dQpOffset[i]);
        }
        else if (mapped)
        {
            std::vector<uint32_t> edgeDensity, avgAngle;
            if (params->aqMode == 4)
            {
                rc = aq_edge_pass(ctx, pic, B, w, h, cols, visited, edgeDensity, avgAngle, wp);
                if (rc)
                    return rc;
            }
            AqFrameParams p;
            p.aqMode = params->aqMode;
            p.depth = ctx->cfg.internalBitDepth;
            p.hdr10 = params->bHDR10Opt != 0;
            p.strength = params->aqStrength;
            p.blockSize = B;
            p.blockCount = blockCount;
            aq_frame_offsets(p, visited, energy.data(), lumaSum.empty() ? nullptr : lumaSum.data(), quantOffsets,
                             qpAqOffset, qpCuTreeOffset, invQscaleFactor, params->aqMode == 4 ? edgeDensity.data() : nullptr,
                             params->aqMode == 4 ? avgAngle.data() : nullptr);
        }
    }
    if (offsets && !mapped)
    {
        // aqMode on with zero strength (:488-505): the caller's offsets alone, or none
        for (int i = 0; i < blockCount; i++)
        {
            const double q = quantOffsets ? quantOffsets[i] : 0.0;
            qpAqOffset[i] = qpCuTreeOffset[i] = q;
            invQscaleFactor[i] = quantOffsets ? aq_exp2fix8(q) : 256;
        }
    }
    if (qg8 && !params->bSkipOffsets && invQscaleFactor && params->invQscaleFactor8x8)
    {
        // :647-660: the lowres 8x8 block's factor is the mean of its four full-resolution 8x8 blocks (also with AQ off,
        // when the arrays exist only for weighted prediction and hold what the caller put there)
        const int W8 = ctx->g.W, H8 = ctx->g.H;
        for (int cuY = 0; cuY < H8; cuY++)
            for (int cuX = 0; cuX < W8; cuX++)
            {
                const int* q = invQscaleFactor + cuX * 2 + cuY * W8 * 4;
                params->invQscaleFactor8x8[cuX + cuY * W8] = (q[0] + q[1] + q[fullCols] + q[fullCols + 1]) / 4;
            }
    }
    int visitW = w, visitH = h;
    if (params->bWeightedPred)
    {
        aq_finish_wp(w, h, wp, wp + 3);
        visitW = ((w + 8) >> 4) << 4;                            // :674-675: the tail below inherits these bounds
        visitH = ((h + 8) >> 4) << 4;
    }
    if (params->bBlockVariance)
    {
        // :688-703: acEnergyCu once more per block (adding to wp_sum / wp_ssd again, as the reference does)
        const int tcols = (visitW + B - 1) / B, trows = (visitH + B - 1) / B;
        std::vector<uint32_t> var((size_t)tcols * trows);
        unsigned long long twp[6];
        if (!staged)
        {
            int rc = stage_aq_picture(ctx, "aq_frame", y, strideY, u, v, strideC, memKind, pic);
            if (rc)
                return rc;
        }
        int rc = aq_energy_pass(ctx, pic, B, visitW, visitH, var.data(), nullptr, twp);
        if (rc)
            return rc;
        for (int i = 0; i < 6; i++)
            wp[i] += twp[i];
        unsigned long long rowVariance = 0;                       // not reset between rows in the reference either
        double frameVariance = 0;
        for (int r = 0; r < trows; r++)
        {
            for (int c = 0; c < tcols; c++)
            {
                params->blockVariance[(size_t)r * tcols + c] = var[(size_t)r * tcols + c];
                rowVariance += var[(size_t)r * tcols + c];
            }
            frameVariance += (double)(rowVariance / (unsigned long long)visitW);
        }
        *params->frameVariance = trows ? frameVariance / visitH : 0.0 / visitH;
    }
    for (int i = 0; i < 3; i++)
    {
        wpSum[i] = wp[i];
        wpSsd[i] = wp[3 + i];
    }
    if (slot >= 0 && offsets)
        return x265gpu_lowres_set_aq(ctx, slot, qg8 ? params->invQscaleFactor8x8 : invQscaleFactor);
    return X265GPU_OK;
}

// Launches the energy pass of every picture, waits for the sums, fills wpSum / wpSsd and hands the double-precision
// mapping of the offsets to host threads.  Returns with those threads running when there is a mapping to do.
int x265gpu_aq_frames_begin(x265gpu_ctx* ctx, int32_t n, x265gpu_aq_picture* pics, int32_t memKind, const x265gpu_aq_params* params)
{
    // reads nothing an intra estimate writes: one that was launched just before keeps running beside the energy pass
    DeviceScope deviceScope(ctx, false);
    if (!ctx || !params || n < 0 || (n && !pics))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    if (ctx->oddSize)
        return fail(ctx, X265GPU_ERR_UNSUPPORTED, "aq_frames: 4:2:0 pictures have even sizes (the context was opened with an odd one)");
    const int w = ctx->cfg.sourceWidth, h = ctx->cfg.sourceHeight;
    const bool offsets = !params->bSkipOffsets && params->aqMode != 0;
    const bool mapped = offsets && params->aqStrength != 0;
    // The batched form covers what a lookahead window normally is: device-resident 4:2:0 pictures, 16x16 AQ blocks,
    // one of the variance modes.  Everything else goes picture by picture through x265gpu_aq_frame.
    const bool fast = memKind == X265GPU_MEM_DEVICE && params->qgSize != 8 && !params->bHevcAq && !params->bBlockVariance &&
                      params->aqMode >= 0 && params->aqMode <= 3 && (mapped || params->bWeightedPred) && (mapped || !offsets);
    if (!fast)
    {
        for (int i = 0; i < n; i++)
        {
            x265gpu_aq_picture& p = pics[i];
            int rc = x265gpu_aq_frame(ctx, p.slot, p.y, p.strideY, p.u, p.v, p.strideC, memKind, params, p.quantOffsets,
                                      p.qpAqOffset, p.qpCuTreeOffset, p.invQscaleFactor, p.wpSum, p.wpSsd);
            if (rc)
                return rc;
        }
        return X265GPU_OK;
    }
    const int cw = w >> 1, ch = h >> 1;
    const int cols = (w + 15) / 16, rows = (h + 15) / 16, blocks = cols * rows;
    for (int i = 0; i < n; i++)
    {
        const x265gpu_aq_picture& p = pics[i];
        if (!p.y || !p.u || !p.v || p.strideY < w || p.strideC < cw || cw < 1 || ch < 1 || (p.slot >= 0 && !slot_ok(ctx, p.slot)) ||
            (mapped && (!p.qpAqOffset || !p.qpCuTreeOffset || !p.invQscaleFactor)))
            return fail(ctx, X265GPU_ERR_ARG, "aq_frames: bad picture");
        if (p.slot >= 0 && blocks != ctx->g.cuCount)
            return fail(ctx, X265GPU_ERR_ARG, "aq_frames: 16x16 block grid differs from the slot's 8x8 lowres grid");
    }
    // one batch at a time shares the context's arrays
    {
        int frc = aq_finish_pending(ctx);
        if (frc)
            return frc;
    }
    // cuTree passes queued so far hold the addresses of what this call rewrites: they run first
    { const int rcTree = cutree_flush(ctx); if (rcTree) return rcTree; }
    const bool hdr = params->bHDR10Opt && mapped;
    const size_t arrayBytes = align256(sizeof(uint32_t) * (size_t)blocks);
    const size_t perPic = (hdr ? 2 : 1) * arrayBytes + AQ_WP_BYTES;
    // after the pictures' arrays: the job structs, then the pointer pairs of the scatter that installs the factors
    const size_t jobsAt = perPic * n, ptrsAt = jobsAt + align256(sizeof(AqJob) * (size_t)n);
    const size_t total = ptrsAt + align256(2 * sizeof(void*) * (size_t)n);
    // The previous batch's factors may still be on their way out of the page-locked arrays (the installing copies
    // are queued behind an intra estimate on the side stream and nobody waits for them on the host): the arrays
    // are not written again before those copies have finished.
    if (ctx->aqInstallInFlight)
    {
        HIPCHK(hipEventSynchronize(ctx->aqInstalled));
        ctx->aqInstallInFlight = false;
    }
    if (total > ctx->aqBufBytes)
    {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->intraStream)
            HIPCHK(hipStreamSynchronize(ctx->intraStream));
        if (ctx->aqDev) (void)hipFree(ctx->aqDev);
        if (ctx->aqHost) (void)hipHostFree(ctx->aqHost);
        ctx->aqDev = ctx->aqHost = nullptr;
        ctx->aqBufBytes = 0;
        if (hipMalloc((void**)&ctx->aqDev, total) != hipSuccess ||
            hipHostMalloc((void**)&ctx->aqHost, total, hipHostMallocDefault) != hipSuccess)
        {
            if (ctx->aqDev) (void)hipFree(ctx->aqDev);
            ctx->aqDev = nullptr;
            return fail(ctx, X265GPU_ERR_NOMEM, "aq_frames: out of memory");
        }
        ctx->aqBufBytes = total;
    }
    x265gpu_ctx::AqPending* b = new (std::nothrow) x265gpu_ctx::AqPending;
    if (!b)
        return X265GPU_ERR_NOMEM;
    b->dev = ctx->aqDev;
    b->host = ctx->aqHost;
    auto drop = [&]() { delete b; };
    // every picture's energy pass in one launch, then one transfer and one wait
    AqJob* hostJobs = (AqJob*)(b->host + jobsAt);
    for (int i = 0; i < n; i++)
    {
        const x265gpu_aq_picture& p = pics[i];
        char* base = b->dev + perPic * i;
        AqJob& job = hostJobs[i];
        memset(&job, 0, sizeof(job));
        job.width = job.visitWidth = w;
        job.height = job.visitHeight = h;
        job.blockSize = 16;
        job.luma = p.y;
        job.u = p.u;
        job.v = p.v;
        job.lumaStride = p.strideY;
        job.chromaStride = p.strideC;
        job.energy = (uint32_t*)base;
        job.lumaSum = hdr ? (uint32_t*)(base + arrayBytes) : nullptr;
        job.wp = (unsigned long long*)(base + (hdr ? 2 : 1) * arrayBytes);
    }
    hipError_t e = hipMemsetAsync(b->dev, 0, perPic * n, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(b->dev + jobsAt, hostJobs, sizeof(AqJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = launch_aq_energy_batch(ctx->stream, ctx->bpp, (const AqJob*)(b->dev + jobsAt), n, w, h);
    if (e == hipSuccess)
        e = hipMemcpyAsync(b->host, b->dev, perPic * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
    {
        drop();
        return fail(ctx, X265GPU_ERR_DEVICE, hipGetErrorString(e));
    }
    const bool wantWp = params->bWeightedPred != 0;
    for (int i = 0; i < n; i++)
    {
        unsigned long long wp[6];
        sum_wp_shards((const unsigned long long*)(b->host + perPic * i + (hdr ? 2 : 1) * arrayBytes), wp);
        if (wantWp)
            aq_finish_wp(w, h, wp, wp + 3);
        for (int k = 0; k < 3; k++)
        {
            pics[i].wpSum[k] = wp[k];
            pics[i].wpSsd[k] = wp[3 + k];
        }
    }
    if (!mapped)
    {
        drop();
        return X265GPU_OK;
    }
    // The double-precision mapping of each picture is independent of the others: host threads share the pictures
    // the way x265's pool workers share the frames of PreLookaheadGroup (slicetype.cpp:1843-1859).
    b->pics.assign(pics, pics + n);
    b->params = *params;
    b->perPic = perPic;
    b->ptrsAt = ptrsAt;
    AqFrameParams ap;
    ap.aqMode = params->aqMode;
    ap.depth = ctx->cfg.internalBitDepth;
    ap.hdr10 = hdr;
    ap.strength = params->aqStrength;
    ap.blockSize = 16;
    ap.blockCount = blocks;
    const char* env = getenv("X265GPU_HOST_THREADS");
    const unsigned hw = std::thread::hardware_concurrency();
    int threads = env ? atoi(env) : (int)(hw ? (hw < 16 ? hw : 16) : 4);
    threads = threads < 1 ? 1 : (threads > n ? n : threads);
    b->next = new std::atomic<int>(0);
    x265gpu_ctx::AqPending* batch = b;
    for (int t = 0; t < threads; t++)
        b->workers.emplace_back([batch, ap, n, perPic, arrayBytes, hdr, blocks]() {
            for (int i = batch->next->fetch_add(1); i < n; i = batch->next->fetch_add(1))
            {
                const x265gpu_aq_picture& p = batch->pics[i];
                const char* base = batch->host + perPic * i;
                aq_frame_offsets(ap, blocks, (const uint32_t*)base, hdr ? (const uint32_t*)(base + arrayBytes) : nullptr,
                                 p.quantOffsets, p.qpAqOffset, p.qpCuTreeOffset, p.invQscaleFactor);
            }
        });
    ctx->aqPending.push_back(b);
    // Job descriptors built from now on point at the slots' factor arrays; the contents are installed by
    // aq_finish_pending, which every pass that reads them goes through before it is launched.
    for (int i = 0; i < n; i++)
        if (pics[i].slot >= 0)
        {
            Slot& s = ctx->slots[pics[i].slot];
            s.hasAq = true;
            if (s.intraFresh)       // estimated before this call: the AQ-weighted sums follow the factors
                ctx->intraDeferred.push_back(pics[i].slot);
        }
    return X265GPU_OK;
}

// Joins the host threads of every batch begun, installs the scale factors for the slots and launches the intra
// estimates that waited for them.
static int aq_finish_pending(x265gpu_ctx* ctx)
{
    if (ctx->aqPending.empty() && ctx->intraDeferred.empty())
        return X265GPU_OK;
    int rc = X265GPU_OK;
    // x265gpu_aq_frames_begin waited for the stream, so the planes these frames' intra estimates read are in place:
    // the uploads and the held-back intra launches depend on nothing queued since and go to the side stream, beside
    // the searches; the main stream joins before its next pass that reads them.
    const bool side = ctx->intraStream && !ctx->prof;
    hipStream_t up = side ? ctx->intraStream : ctx->stream;
    if (ctx->aqInstalledOnMain)
    {
        // factors uploaded on the main stream (x265gpu_lowres_set_aq) must be in place before the sums are redone
        if (side && (hipEventRecord(ctx->intraFork, ctx->stream) != hipSuccess ||
                     hipStreamWaitEvent(up, ctx->intraFork, 0) != hipSuccess))
            rc = X265GPU_ERR_DEVICE;
        ctx->aqInstalledOnMain = false;
    }
    std::vector<x265gpu_ctx::AqPending*> batches;
    batches.swap(ctx->aqPending);
    for (x265gpu_ctx::AqPending* b : batches)
    {
        for (std::thread& th : b->workers)
            th.join();
        // The scale factors of every picture that has a slot: into the page-locked arrays (each picture's energy
        // array has served its purpose and is exactly as large), one transfer, one pass that hands them to the slots.
        const size_t bytes = sizeof(int32_t) * ctx->g.cuCount;
        const int32_t** src = (const int32_t**)(b->host + b->ptrsAt);
        int32_t** dst = (int32_t**)(b->host + b->ptrsAt) + b->pics.size();
        int count = 0;
        for (size_t i = 0; i < b->pics.size(); i++)
        {
            const x265gpu_aq_picture& p = b->pics[i];
            if (p.slot < 0)
                continue;
            memcpy(b->host + b->perPic * i, p.invQscaleFactor, bytes);
            src[count] = (const int32_t*)(b->dev + b->perPic * i);
            dst[count] = ctx->slots[p.slot].invQscale;
            ctx->slots[p.slot].hasAq = true;
            count++;
        }
        if (count && rc == X265GPU_OK)
        {
            // pointer pairs packed behind each other for the device: count sources, then count destinations
            memmove((void*)(src + count), (const void*)dst, sizeof(void*) * (size_t)count);
            hipError_t e = hipMemcpyAsync(b->dev, b->host, b->perPic * b->pics.size(), hipMemcpyHostToDevice, up);
            if (e == hipSuccess)
                e = hipMemcpyAsync(b->dev + b->ptrsAt, b->host + b->ptrsAt, 2 * sizeof(void*) * (size_t)count, hipMemcpyHostToDevice, up);
            if (e == hipSuccess)
                e = launch_scatter_i32(up, (const int32_t* const*)(b->dev + b->ptrsAt),
                                       (int32_t* const*)(b->dev + b->ptrsAt) + count, (int)ctx->g.cuCount, count);
            if (e == hipSuccess)
                e = hipEventRecord(ctx->aqInstalled, up);
            if (e != hipSuccess)
                rc = X265GPU_ERR_DEVICE;
            else
                ctx->aqInstallInFlight = true;
        }
        delete b->next;
        delete b;
    }
    if (rc == X265GPU_OK && !ctx->intraDeferred.empty())
    {
        // the intra estimates launched while the factors were still being computed: their AQ-weighted sums, redone
        // behind the estimate (same stream) and behind the factors' installation just queued
        std::vector<int32_t> slots;
        slots.swap(ctx->intraDeferred);
        std::sort(slots.begin(), slots.end());
        slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
        std::vector<IntraJob> jobs(slots.size());
        for (size_t i = 0; i < slots.size(); i++)
        {
            Slot& s = ctx->slots[slots[i]];
            memset(&jobs[i], 0, sizeof(IntraJob));
            jobs[i].invQscale = s.hasAq ? s.invQscale : nullptr;
            jobs[i].intraCost = s.intraCost;
            jobs[i].rowSatds = s.rowSatds;
            jobs[i].rowPartials = s.rowPartials;
            jobs[i].sums = s.sums;
        }
        const size_t bytes = sizeof(IntraJob) * jobs.size();
        rc = ensure_intra_job_buf(ctx, bytes);
        if (rc == X265GPU_OK)
        {
            char* host = ctx->intraJobHost + ctx->intraJobBufSize;
            char* dev = ctx->intraJobBuf + ctx->intraJobBufSize;
            memcpy(host, jobs.data(), bytes);
            hipError_t e = hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, up);
            if (e == hipSuccess)
                e = launch_intra_aq_sums(up, ctx->g, (const IntraJob*)dev, (int)jobs.size());
            if (e == hipSuccess)
                e = hipEventRecord(ctx->intraDone, up);
            if (e != hipSuccess)
                rc = X265GPU_ERR_DEVICE;
            else if (side)
                ctx->intraPending = true;
        }
    }
    else if (rc == X265GPU_OK && side && !batches.empty())
    {
        if (hipEventRecord(ctx->intraDone, up) != hipSuccess)
            rc = X265GPU_ERR_DEVICE;
        ctx->intraPending = true;
    }
    ctx->intraDeferred.clear();
    join_intra(ctx);
    return rc;
}

int x265gpu_aq_motion_frame(x265gpu_ctx* ctx, int32_t slot, int32_t d0, int32_t d1, double* qpAqOffset,
                            double* qpCuTreeOffset, int32_t* invQscaleFactor, double* qpAqMotionOffset)
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);          // the arrays below are complete, and in the slot, before they are added to
    if (!ctx || !slot_ok(ctx, slot) || d0 < 0 || d0 >= ctx->D || d1 < 0 || d1 >= ctx->D || !qpAqOffset || !qpCuTreeOffset ||
        !invQscaleFactor || !qpAqMotionOffset)
        return X265GPU_ERR_ARG;
    const int n = ctx->g.cuCount, D = ctx->D;
    if (!ctx->motionDev)
    {
        HIPCHK(hipMalloc((void**)&ctx->motionDev, sizeof(double) * (size_t)n));
        HIPCHK(hipHostMalloc((void**)&ctx->motionHost, sizeof(double) * (size_t)n, hipHostMallocDefault));
    }
    Slot& sb = ctx->slots[slot];
    HIPCHK(launch_aq_motion(ctx->stream, sb.lowresCosts + ((size_t)d0 * D + d1) * n, sb.mvs + ((size_t)(0 * D + d0) * n) * 2,
                            sb.mvs + ((size_t)(1 * D + d1) * n) * 2, n, ctx->motionDev));
    HIPCHK(hipMemcpyAsync(ctx->motionHost, ctx->motionDev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (aq_motion_offsets(n, ctx->motionHost, qpAqMotionOffset, qpAqOffset, qpCuTreeOffset, invQscaleFactor) == 0)
        return X265GPU_OK;
    // The raised factors go to the slot for the cost passes and cuTree passes that follow.  The picture's intra
    // estimate keeps the sums it formed with the factors of its time, as in x265, where lowresIntraEstimate has
    // long run when aqMotion changes invQscaleFactor.
    return set_aq_staged(ctx, slot, invQscaleFactor, nullptr, false);
}

void x265gpu_aq_memo_stats(uint64_t* lookups, uint64_t* hits)
{
    aq_activity_memo_stats(lookups, hits);
}

int x265gpu_aq_frames_finish(x265gpu_ctx* ctx)
{
    DeviceScope deviceScope(ctx);
    if (!ctx)
        return X265GPU_ERR_ARG;
    FINISH_AQ(ctx);
    return X265GPU_OK;
}

int x265gpu_aq_frames(x265gpu_ctx* ctx, int32_t n, x265gpu_aq_picture* pics, int32_t memKind, const x265gpu_aq_params* params)
{
    int rc = x265gpu_aq_frames_begin(ctx, n, pics, memKind, params);
    if (rc == X265GPU_OK && ctx)
        rc = x265gpu_aq_frames_finish(ctx);
    return rc;
}

int x265gpu_get_frame_costs(x265gpu_ctx* ctx, int32_t n, const int32_t* slots, const int32_t* d0, const int32_t* d1,
                            x265gpu_frame_cost* out)
{
    DeviceScope deviceScope(ctx);
    FINISH_AQ(ctx);
    if (!ctx || n < 0 || (n && (!slots || !out)))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    std::vector<long long> sums((size_t)3 * n);
    std::vector<const long long*> where((size_t)n);
    for (int i = 0; i < n; i++)
    {
        const int a = d0 ? d0[i] : 0, b = d1 ? d1[i] : 0;
        if (!slot_ok(ctx, slots[i]) || a < 0 || a >= ctx->D || b < 0 || b >= ctx->D)
            return X265GPU_ERR_ARG;
        where[i] = ctx->slots[slots[i]].sums + ((size_t)a * ctx->D + b) * 3;
    }
    if (n == 1)
        HIPCHK(hipMemcpyAsync(sums.data(), where[0], 3 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    else
    {
        // the totals sit in n different slots: gathered on the device, one transfer each way
        const size_t ptrBytes = align256(sizeof(long long*) * (size_t)n);
        int rc = ensure_pu_buf(ctx, ptrBytes + 3 * sizeof(long long) * (size_t)n);
        if (rc)
            return rc;
        HIPCHK(hipMemcpyAsync(ctx->puBuf, where.data(), sizeof(long long*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(launch_gather_totals(ctx->stream, (const long long* const*)ctx->puBuf, (long long*)(ctx->puBuf + ptrBytes), n));
        HIPCHK(hipMemcpyAsync(sums.data(), ctx->puBuf + ptrBytes, 3 * sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; i++)
    {
        const int a = d0 ? d0[i] : 0, b = d1 ? d1[i] : 0;
        long long score = sums[3 * (size_t)i];
        if (b != 0)
            score = score * 100 / (130 + ctx->cfg.bFrameBias);
        out[i].costEst = score;
        out[i].costEstAq = sums[3 * (size_t)i + 1];
        out[i].intraMbs = (a == 0 && b == 0) ? 0 : (int32_t)sums[3 * (size_t)i + 2];
        out[i].reserved = 0;
    }
    return X265GPU_OK;
}

int x265gpu_hist_stats(x265gpu_ctx* ctx, int32_t slot, const void* y, intptr_t strideY, const void* u, const void* v,
                       intptr_t strideC, int32_t memKind, x265gpu_hist_stats_t* out)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slot) || !y || !u || !v || !out)
        return X265GPU_ERR_ARG;
    if (ctx->oddSize)
        return fail(ctx, X265GPU_ERR_UNSUPPORTED, "hist_stats: 4:2:0 pictures have even sizes (the context was opened with an odd one)");
    const int w = ctx->cfg.sourceWidth, h = ctx->cfg.sourceHeight;
    if (w < 64 || h < 64 || strideY < w || strideC < (w >> 1))
        return fail(ctx, X265GPU_ERR_ARG, "hist_stats: picture smaller than 64x64 or bad stride");
    const int maxRows = (h + 7) / 8;
    const size_t histBytes = sizeof(uint32_t) * 16 * 3 * 256, sumBytes = sizeof(unsigned long long) * 16 * 3;
    const size_t rowBytes = sizeof(unsigned long long) * 3 * (size_t)maxRows;
    const size_t total = align256(histBytes) + align256(sumBytes) + align256(rowBytes);
    int rc = ensure_pu_buf(ctx, total);
    if (rc)
        return rc;
    YuvOnDevice pic;
    rc = stage_yuv(ctx, y, strideY, u, v, strideC, memKind, pic);
    if (rc)
        return rc;
    HistJob job;
    memset(&job, 0, sizeof(job));
    job.lowres0 = ctx->slots[slot].planes + (size_t)ctx->g.padOffset * ctx->bpp;
    job.lowresStride = ctx->g.stride;
    job.y = pic.y;
    job.u = pic.u;
    job.v = pic.v;
    job.strideY = pic.strideY;
    job.strideC = pic.strideC;
    job.width = w;
    job.height = h;
    job.depth = ctx->cfg.internalBitDepth;
    job.hist = (uint32_t*)ctx->puBuf;
    job.sums = (unsigned long long*)(ctx->puBuf + align256(histBytes));
    job.rowVar = (unsigned long long*)(ctx->puBuf + align256(histBytes) + align256(sumBytes));
    job.maxRows = maxRows;
    HIPCHK(hipMemsetAsync(ctx->puBuf, 0, total, ctx->stream));
    HIPCHK(launch_hist_stats(ctx->stream, ctx->bpp, job));
    std::vector<uint32_t> hist(16 * 3 * 256);
    std::vector<unsigned long long> sums(16 * 3), rowVar(3 * (size_t)maxRows);
    HIPCHK(hipMemcpyAsync(hist.data(), job.hist, histBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(sums.data(), job.sums, sumBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(rowVar.data(), job.rowVar, rowBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    hist_finish(w, h, hist.data(), sums.data(), rowVar.data(), maxRows, out);
    return X265GPU_OK;
}

int x265gpu_bipred_satd(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef0, int32_t slotRef1, int32_t n,
                        const x265gpu_bipred_job* jobs, int32_t* costs)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || n < 0 || (n && (!jobs || !costs)))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    if (!slot_ok(ctx, slotFenc) || !slot_ok(ctx, slotRef0) || !slot_ok(ctx, slotRef1))
        return fail(ctx, X265GPU_ERR_ARG, "bipred_satd: bad slot");
    if (!ctx->fullStride)
        return fail(ctx, X265GPU_ERR_STATE, "bipred_satd: context opened without bKeepFullRes");
    PuFrame f = pu_frame(ctx, slotFenc, slotRef0);
    for (int i = 0; i < n; i++)
    {
        const x265gpu_bipred_job& j = jobs[i];
        bool ok = j.w >= 4 && j.h >= 4 && j.w <= 64 && j.h <= 64 && !(j.w & 3) && !(j.h & 3) && j.x >= 0 && j.y >= 0 &&
                  j.x + j.w <= ctx->fullW && j.y + j.h <= ctx->fullH;
        for (int k = 0; k < 2 && ok; k++)
        {
            const int fx = j.x + ((k ? j.qmv1X : j.qmv0X) >> 2), fy = j.y + ((k ? j.qmv1Y : j.qmv0Y) >> 2);
            ok = fx >= -ctx->fullMarginX + 8 && fx + j.w <= ctx->fullW + ctx->fullMarginX - 16 &&
                 fy >= -ctx->fullMarginY + 8 && fy + j.h <= ctx->fullH + ctx->fullMarginY - 8;
        }
        if (!ok)
            return fail(ctx, X265GPU_ERR_ARG, "bipred_satd: bad job");
        const int segW = (j.w & 7) ? 4 : 8;
        const int passes = ((j.w / segW) * (j.h / 4) + 15) / 16;
        f.maxPasses = passes > f.maxPasses ? passes : f.maxPasses;
    }
    const size_t jobBytes = align256(sizeof(x265gpu_bipred_job) * (size_t)n), outBytes = sizeof(int32_t) * (size_t)n;
    int rc = ensure_pu_buf(ctx, jobBytes + outBytes);
    if (rc)
        return rc;
    const size_t origin = ((size_t)ctx->fullMarginY * ctx->fullStride + ctx->fullMarginX) * ctx->bpp;
    HIPCHK(hipMemcpyAsync(ctx->puBuf, jobs, sizeof(x265gpu_bipred_job) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(launch_bipred_satd(ctx->stream, ctx->bpp, f, ctx->slots[slotRef1].full + origin, ctx->puBuf, n,
                              (int*)(ctx->puBuf + jobBytes)));
    HIPCHK(hipMemcpyAsync(costs, ctx->puBuf + jobBytes, outBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_block_compare(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef, int32_t n, const x265gpu_block_job* jobs,
                          int32_t* costs)
{
    DeviceScope deviceScope(ctx);
    static_assert(sizeof(x265gpu_block_job) == sizeof(BlockJob), "public and device job layouts must match");
    if (!ctx || n < 0 || (n && (!jobs || !costs)))
        return X265GPU_ERR_ARG;
    if (!n)
        return X265GPU_OK;
    if (!slot_ok(ctx, slotFenc) || !slot_ok(ctx, slotRef))
        return fail(ctx, X265GPU_ERR_ARG, "block_compare: bad slot");
    if (!ctx->fullStride)
        return fail(ctx, X265GPU_ERR_STATE, "block_compare: context opened without bKeepFullRes");
    PuFrame f = pu_frame(ctx, slotFenc, slotRef);
    for (int i = 0; i < n; i++)
    {
        const x265gpu_block_job& j = jobs[i];
        const int fx = j.x + (j.qmvX >> 2), fy = j.y + (j.qmvY >> 2);
        bool ok = j.w >= 4 && j.h >= 4 && j.w <= 64 && j.h <= 64 && !(j.w & 3) && !(j.h & 3) && j.x >= 0 && j.y >= 0 &&
                  j.x + j.w <= ctx->fullW && j.y + j.h <= ctx->fullH && j.op >= X265GPU_OP_SAD && j.op <= X265GPU_OP_SA8D &&
                  fx >= -ctx->fullMarginX + 8 && fx + j.w <= ctx->fullW + ctx->fullMarginX - 16 &&
                  fy >= -ctx->fullMarginY + 8 && fy + j.h <= ctx->fullH + ctx->fullMarginY - 8;
        if (ok && j.op == X265GPU_OP_SA8D)
            ok = j.w == j.h && (j.w == 8 || j.w == 16 || j.w == 32 || j.w == 64) && !((j.qmvX | j.qmvY) & 3);
        if (!ok)
            return fail(ctx, X265GPU_ERR_ARG, "block_compare: bad job");
        const int segW = (j.w & 7) ? 4 : 8;
        const int passes = ((j.w / segW) * (j.h / 4) + 15) / 16;
        f.maxPasses = passes > f.maxPasses ? passes : f.maxPasses;
    }
    const size_t jobBytes = align256(sizeof(BlockJob) * (size_t)n), outBytes = sizeof(int32_t) * (size_t)n;
    int rc = ensure_pu_buf(ctx, jobBytes + outBytes);
    if (rc)
        return rc;
    HIPCHK(hipMemcpyAsync(ctx->puBuf, jobs, sizeof(BlockJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(launch_block_compare(ctx->stream, ctx->bpp, f, (const BlockJob*)ctx->puBuf, n, (int*)(ctx->puBuf + jobBytes)));
    HIPCHK(hipMemcpyAsync(costs, ctx->puBuf + jobBytes, outBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_interp_luma(x265gpu_ctx* ctx, int32_t slot, int32_t x, int32_t y, int32_t w, int32_t h, int32_t xFrac,
                        int32_t yFrac, void* out)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !out || !slot_ok(ctx, slot))
        return X265GPU_ERR_ARG;
    if (!ctx->fullStride)
        return fail(ctx, X265GPU_ERR_STATE, "interp_luma: context opened without bKeepFullRes");
    if (w < 1 || h < 1 || w > 64 || h > 64 || xFrac < 0 || xFrac > 3 || yFrac < 0 || yFrac > 3 ||
        x < -ctx->fullMarginX + 8 || x + w > ctx->fullW + ctx->fullMarginX - 16 ||
        y < -ctx->fullMarginY + 8 || y + h > ctx->fullH + ctx->fullMarginY - 8)
        return fail(ctx, X265GPU_ERR_ARG, "interp_luma: bad block");
    const size_t bytes = (size_t)w * h * ctx->bpp;
    int rc = ensure_pu_buf(ctx, bytes);
    if (rc)
        return rc;
    const PuFrame f = pu_frame(ctx, slot, slot);
    HIPCHK(launch_interp_block(ctx->stream, ctx->bpp, f.ref, f.stride, x, y, w, h, xFrac, yFrac, f.depth, ctx->puBuf));
    HIPCHK(hipMemcpyAsync(out, ctx->puBuf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return X265GPU_OK;
}

int x265gpu_get_full_plane(x265gpu_ctx* ctx, int32_t slot, void* out, int32_t* stride, int32_t* rows, int32_t* originX,
                           int32_t* originY)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || !slot_ok(ctx, slot))
        return X265GPU_ERR_ARG;
    if (!ctx->fullStride)
        return fail(ctx, X265GPU_ERR_STATE, "get_full_plane: context opened without bKeepFullRes");
    if (stride) *stride = ctx->fullStride;
    if (rows) *rows = ctx->fullRows;
    if (originX) *originX = ctx->fullMarginX;
    if (originY) *originY = ctx->fullMarginY;
    if (out)
    {
        HIPCHK(hipMemcpyAsync(out, ctx->slots[slot].full, (size_t)ctx->fullStride * ctx->fullRows * ctx->bpp,
                              hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return X265GPU_OK;
}

int x265gpu_profile_enable(x265gpu_ctx* ctx, int32_t on)
{
    DeviceScope deviceScope(ctx);
    if (!ctx)
        return X265GPU_ERR_ARG;
    ctx->prof = on != 0;
    return X265GPU_OK;
}

int x265gpu_profile_read(x265gpu_ctx* ctx, int32_t phase, double* ms, int64_t* launches, int32_t reset)
{
    DeviceScope deviceScope(ctx);
    if (!ctx || phase < 0 || phase >= X265GPU_PHASE_COUNT)
        return X265GPU_ERR_ARG;
    if (ms)
        *ms = ctx->profMs[phase];
    if (launches)
        *launches = ctx->profLaunches[phase];
    if (reset)
    {
        ctx->profMs[phase] = 0;
        ctx->profLaunches[phase] = 0;
    }
    return X265GPU_OK;
}

} // extern "C"
