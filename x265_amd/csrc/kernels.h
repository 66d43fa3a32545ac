// Job descriptors passed from the host-side C-ABI to the kernels, and the
// launch entry points each .hip file exports.  Pointers in a job are already
// resolved to device addresses of one slot's arrays, so kernels never chase a
// slot table.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xg {

struct Geometry
{
    int srcWidth, srcHeight;
    int width, lines, stride;       // lowres, rounded to 8; stride in pixels
    int marginX, marginY;
    int W, H, cuCount;              // 8x8 block grid of the half-res planes
    int W4, H4, cuCount4;           // block grid of the quarter-res planes (HME level 0)
    long long planeSize, padOffset; // pixels
    long long planeSize4, padOffset4;
    int stride4, width4, lines4, marginX4, marginY4;
    int mcstfFilter;                // 1: (a+b+c+d+2)>>2 downscale (frameInitLowresCoreMCSTF, source/common/pixel.cpp:1017-1039)
};

struct LowresJob
{
    const void* src;      // full-res luma, pixel (0,0)
    long long srcStride;  // pixels
    void* planes;         // buffer start of the 4 half-res planes
    void* planes4;        // buffer start of the 4 quarter-res planes or null
};

struct IntraJob
{
    const void* plane0;       // origin of half-res plane 0
    const int32_t* invQscale; // or null
    int32_t* intraCost;
    uint8_t* intraMode;
    uint16_t* lowresCosts;    // lowresCosts[0][0]
    int32_t* rowSatds;        // rowSatds[0][0]
    long long* rowPartials;   // [H][2]: costEst, costEstAq partial per block row
    long long* sums;          // [2] frame totals costEst[0][0], costEstAq[0][0], written by the finalize pass
};

struct SearchJob
{
    const void* fencPlane0;   // origin of the encode frame's plane 0 (of the searched resolution)
    const void* refOrigin;    // origin of plane 0 of the 4-plane reference set that is searched
    const void* zeroRef;      // origin of the plane the closing zero-vector check reads
    int zeroStride;           // and its stride
    int32_t* mvs;             // {x,y} per block, this (frame, list, distance)
    int32_t* mvCosts;
    const int32_t* hmeMvs;    // quarter-res vectors used as an extra predictor, or null
    const int32_t* hmeMvCosts;
    int bBidir;
    int level0;               // 1: this job searches the quarter-res planes
    int method;
    int merange;
    int numSlices;            // cooperative slices this search is split into (1: whole frame)
    int sliceRows;            // block rows per slice; the last slice also takes the remainder
};

struct CostJob
{
    const void* fencPlane0;
    const void* ref0Origin;
    const void* ref1Origin;
    const int32_t* mvs0;
    const int32_t* mvs1;
    const int32_t* mvCosts0;
    const int32_t* mvCosts1;
    const int32_t* intraCost;
    const int32_t* invQscale;   // or null
    uint16_t* lowresCosts;
    int32_t* rowSatds;
    long long* rowPartials;     // [H][3]: costEst, costEstAq, intraMbs per block row
    long long* sums;            // [3] frame totals written by the finalize pass
    long long* batchOut;        // [3] second copy in a per-batch contiguous array, or null
    int bBidir;
};

struct WeightCostJob
{
    const void* fencPlane0;
    const void* refPlane0;      // origin of (possibly weighted) plane 0
    const int32_t* intraCost;
    unsigned int* out;          // single accumulator, zeroed by the host beforehand
    // weighted != 0: refPlane0 is the unweighted reference and weight_pp (w0, round, shift, offset; `correction`
    // = 14 - depth, clip to pixelMax) is applied to each row as it is read, instead of to a copy of the plane
    int weighted, w0, round, shift, offset, correction, pixelMax, pad;
};

// launchers (defined in the .hip files); bytesPerPixel is 1 or 2
hipError_t launch_gather_totals(hipStream_t s, const long long* const* dSrc, long long* dDst, int n);
hipError_t launch_weight_cost_batch(hipStream_t s, const Geometry& g, int bpp, const WeightCostJob* dJobs, int n);
hipError_t launch_lowres_init(hipStream_t s, const Geometry& g, int bpp, const LowresJob* dJobs, int n);
hipError_t launch_intra(hipStream_t s, const Geometry& g, int bpp, int depth, int lambda, const IntraJob* dJobs, int n);
// redo the AQ-weighted row and frame sums of finished intra estimates with the jobs' current scale factors
hipError_t launch_intra_aq_sums(hipStream_t s, const Geometry& g, const IntraJob* dJobs, int n);
// Side streams the search may spread independent jobs over so that the drain of
// one anti-diagonal launch overlaps the body of another job group's launch.
struct SearchStreams
{
    int count = 0;                 // side streams in use (0: everything on the main stream)
    hipStream_t side[3] = { nullptr, nullptr, nullptr };
    hipEvent_t fork = nullptr;     // recorded on the main stream, awaited by the side streams
    hipEvent_t join[3] = { nullptr, nullptr, nullptr };
};
// slicesMax / sliceHeightMax: the largest numSlices and the tallest slice (in block rows) over the n jobs
hipError_t launch_search(hipStream_t s, const SearchStreams& aux, const Geometry& g, int bpp, const SearchJob* dJobs, int n,
                         int level0, int method, const uint16_t* costTab, int maxMvd, int slicesMax, int sliceHeightMax,
                         int* launches);
// single-launch search (row items with in-launch hand-off); dSync holds search_rows_sync_bytes() bytes,
// word SEARCH_SYNC_ERROR of it is a sticky error flag (a wait ran out) that the caller zeroes once and reads back
constexpr int SEARCH_SYNC_ERROR = 64;
size_t search_rows_sync_bytes(const Geometry& g, int n);
hipError_t launch_search_rows(hipStream_t s, const Geometry& g, int bpp, const SearchJob* dJobs, int n, int level0,
                              int method, const uint16_t* costTab, int maxMvd, int* dSync, int* launches);
// single-launch search in diagonal order with the vectors as ready flags (search_kernels.hip); the chunk table
// lists (diagonal, first block) of every group of four blocks: search_flow_chunks returns its length and fills it
int search_flow_chunks(int W, int H, uint32_t* out);
hipError_t launch_search_flow(hipStream_t s, const Geometry& g, int bpp, const SearchJob* dJobs, int n, int level0, int method,
                              const uint32_t* dChunkTab, int numChunks, const uint16_t* costTab, int maxMvd, int* dSync,
                              int gridPad, int* launches);
// ---- full-resolution PU motion estimation ----
// same layout as x265gpu_pu_job / x265gpu_pu_result
struct PuJob
{
    int x, y, w, h;                    // PU rectangle in the picture, pixels
    int mvpx, mvpy;                    // quarter-pel predictor
    int minx, miny, maxx, maxy;        // full-pel search bounds
    int numCandidates;                 // extra quarter-pel candidates (at most 12)
    int mvc[24];
    int flags;                         // bit 0: m_vertRestriction
};
struct PuResult { int mvx, mvy, cost, reserved; };
struct PuFrame
{
    const void* fenc;                  // pixel (0,0) of the padded encode picture
    const void* ref;                   // pixel (0,0) of the padded reference picture
    int stride;
    int width, height;                 // interior of the padded pictures (multiples of the CTU size)
    int marginX, marginY;
    int depth;
    int method, merange, subpelRefine;
    int maxPasses;                     // largest ceil(tiles / 16) over the batch: sizes the LDS per wave
    int rowForm;                       // 1: SAD stages read the encode block in row form (pu_kernels.hip)
    const uint32_t* seaSums[12];       // successive elimination: entry (0,0) of the reference's block-sum planes
};
struct BlockJob { int x, y, w, h; int qmvx, qmvy; int op; int reserved; };   // same layout as x265gpu_block_job
hipError_t launch_block_compare(hipStream_t s, int bpp, const PuFrame& f, const BlockJob* dJobs, int n, int* dOut);
// SATD against the average of two predictions (x265gpu_bipred_job layout); ref1 = pixel (0,0) of the second padded reference
hipError_t launch_bipred_satd(hipStream_t s, int bpp, const PuFrame& f, const void* ref1, const void* dJobs, int n, int* dOut);
hipError_t launch_interp_block(hipStream_t s, int bpp, const void* pic, int stride, int x, int y, int w, int h, int xf, int yf,
                               int depth, void* dOut);
// block sums of a padded picture for the planes in `mask` (sea_shape.h); tmp holds stride * rows entries
hipError_t launch_sea_planes(hipStream_t s, int bpp, const void* pic, int stride, int rows, unsigned mask, uint32_t* tmp,
                             uint32_t* const planes[12]);
hipError_t launch_pad_picture(hipStream_t s, int bpp, const void* src, long long srcStride, void* dst, int stride, int rows,
                              int width, int height, int marginX, int marginY);
// Jobs sorted into size classes: order[] lists job indices class after class (PUs of at most 2, 4, 8 tiles of
// 8x4 / 4x4 pixels share a wave eight, four, two at a time; the rest own a wave each: those of at most 32 tiles,
// then the larger ones), count[] the class sizes.
struct PuClasses
{
    const int* order;      // device
    int count[5];
    int maxPasses[2];      // largest ceil(tiles / 16) in each of the two wave-per-PU classes
};
// the shared-wave classes run on aux.side[0] beside the one-PU-per-wave class on s (when aux has a side stream)
hipError_t launch_pu_search(hipStream_t s, const SearchStreams& aux, int bpp, const PuFrame& f, const PuJob* dJobs,
                            const PuClasses& classes, const uint16_t* costTab, int maxMvd, PuResult* dOut);

// AQ block energies: tight (unpadded) planes on the device, 4:2:0
struct AqJob
{
    const void* luma; const void* u; const void* v;
    long long lumaStride, chromaStride;      // pixels
    int width, height;                       // luma size
    int visitWidth, visitHeight;             // the block loop runs while blockX < visitWidth, blockY < visitHeight
    int blockSize;                           // 16 (qgSize >= 16) or 8 (qgSize 8)
    uint32_t* energy;                        // [ceil(visitHeight/blockSize)][ceil(visitWidth/blockSize)]
    uint32_t* lumaSum;                       // same shape: sum of the block's luma samples (lumaSumCu), or null
    unsigned long long* wp;                  // AQ_WP_SHARDS sets of { wp_sum Y,U,V, wp_ssd Y,U,V }, AQ_WP_SHARD_STRIDE words
                                             // apart, zeroed by the caller; the frame sums are the totals over the shards
};
constexpr int AQ_WP_SHARDS = 64, AQ_WP_SHARD_STRIDE = 16;    // one 128-byte line per shard
constexpr size_t AQ_WP_BYTES = (size_t)AQ_WP_SHARDS * AQ_WP_SHARD_STRIDE * sizeof(unsigned long long);
hipError_t launch_aq_energy(hipStream_t s, int bpp, const AqJob& job);
hipError_t launch_aq_energy_batch(hipStream_t s, int bpp, const AqJob* dJobs, int n, int visitWidth, int visitHeight);
// --aq-motion: per-block vector length of a request's decision (aq_kernels.hip), then the host half (aq_host.cpp):
// calcMotionAdaptiveQuantFrame from the displacements on; returns the number of blocks whose offsets changed
hipError_t launch_aq_motion(hipStream_t s, const uint16_t* lowresCosts, const int32_t* mvs0, const int32_t* mvs1, int n,
                            double* displacement);
int aq_motion_offsets(int blocks, const double* displacement, double* qpAqMotionOffset, double* qpAqOffset,
                      double* qpCuTreeOffset, int32_t* invQscale);
hipError_t launch_scatter_i32(hipStream_t s, const int32_t* const* dSrc, int32_t* const* dDst, int count, int n);
// xPreanalyze's quadrant sums: out holds 8 values per part (4 sums, 4 sums of squares), parts in raster order
hipError_t launch_aq_quadrants(hipStream_t s, int bpp, const void* luma, long long stride, int width, int height, int partW,
                               int partH, unsigned long long* out);
// xPreanalyze / xPreanalyzeQp for one layer from those sums (aq_host.cpp); returns dAvgActivity
double hevc_aq_layer(int width, int height, int partW, int partH, int numPartInWidth, int numPartInHeight,
                     double qpAdaptationRange, const unsigned long long* quadrants, double* dActivity, double* dQpOffset,
                     double* dCuTreeOffset);

// Host half of calcAdaptiveQuantFrame: block energies -> QP offsets (aq_host.cpp).
// The edge mode's picture pass (X265_AQ_EDGE): per AQ block the sum and the sum of squares of the edge picture and
// the sum of the per-pixel gradient angles.  A pixel whose angle in whole degrees cannot be told apart from its
// neighbour within the accuracy of the device's atan2f is left out of the sum and listed instead, for the host to
// add with its own atan2f (edge_theta).
struct AqEdgeJob
{
    const void* luma;
    long long lumaStride;                    // pixels
    int width, height;
    int blockSize;                           // 16 or 8
    int blocksX;
    int tileRow0;                            // first row of 16x16 tiles this launch covers
    int theta180, theta90pos, theta90neg;    // the host's angles for gradients along the axes
    uint32_t* edgeSum;                       // [blocks], zeroed by the caller
    uint32_t* edgeSsd;
    uint32_t* thetaSum;
    uint32_t* listCount;                     // one counter, zeroed per launch
    uint32_t* list;                          // triples { block index, gradientV, gradientH }, capacity = pixels of the launch
};
hipError_t launch_aq_edge(hipStream_t s, int bpp, int depth, const AqEdgeJob& job, int tileRows);
int edge_theta(int gradientV, int gradientH);    // aq_host.cpp

struct AqFrameParams
{
    int aqMode;          // 1 variance, 2 auto-variance, 3 auto-variance biased, 4 edge
    int depth;
    int hdr10;
    double strength;     // rc.aqStrength
    int blockSize;       // 16, or 8 for qgSize 8
    int blockCount;      // what the activity moments are divided by (can exceed the visited blocks with qgSize 8)
};
int aq_exp2fix8(double x);
// edgeDensity / avgAngle: per block, edge mode only (else null)
void aq_frame_offsets(const AqFrameParams& p, int blocks, const uint32_t* energy, const uint32_t* lumaSum,
                      const float* quantOffsets, double* qpAqOffset, double* qpCuTreeOffset, int32_t* invQscale,
                      const uint32_t* edgeDensity = nullptr, const uint32_t* avgAngle = nullptr);
// lookups of the per-energy activity table behind aq_frame_offsets, and how many it answered (process-wide, since start)
void aq_activity_memo_stats(uint64_t* lookups, uint64_t* hits);
void aq_finish_wp(int width, int height, unsigned long long sum[3], unsigned long long ssd[3]);
// cuTreeFinish for qgSize > 8 (aq_host.cpp): propagate totals -> qpCuTreeOffset
void cutree_finish_offsets(int blocks, const int32_t* intraCost, const int32_t* invQscale, const unsigned long long* totals,
                           int fpsFactor, double weightDelta, double strength, const double* qpAqOffset,
                           double* qpCuTreeOffset);
void cutree_finish_hevc(int W, int H, int widthFullRes, int heightFullRes, bool qg8, const int32_t* intraCost,
                        const int32_t* invQscale, const unsigned long long* totals, int fpsFactor, double weightDelta,
                        double strength, int numLayers, const int* partW, const int* partH, const int* numW, const int* numH,
                        const double* const* dQpOffset, double* const* dCuTreeOffset, double* ratio8x8);
void cutree_finish_offsets_qg8(int W, int H, const int32_t* intraCost, const int32_t* invQscale8x8,
                               const unsigned long long* totals, int fpsFactor, double weightDelta, double strength,
                               const double* qpAqOffset, double* qpCuTreeOffset);

// One estimateCUPropagate(p0, p1, b): frame b's blocks hand their amounts to frames p0 / p1.
struct PropagateJob
{
    const int32_t* intraCost;
    const uint16_t* lowresCosts;             // lowresCosts[b - p0][p1 - b]
    const int32_t* invQscale;
    const unsigned long long* propagateIn;   // frame b's own totals, or null for an unreferenced frame
    const int32_t* mvs[2];                   // lowresMvs[0][b - p0], lowresMvs[1][p1 - b]
    unsigned long long* refCosts[2];         // totals of frames p0 and p1
    int W, H;
    int bipredWeight;                        // list 0's share of 64
    double fps;                              // fpsFactor / 256
};
hipError_t launch_cutree_propagate(hipStream_t s, const PropagateJob& job);
hipError_t launch_cutree_propagate_batch(hipStream_t s, const PropagateJob* dJobs, int count, int W, int H);
hipError_t launch_cutree_clear_batch(hipStream_t s, unsigned long long* const* dCells, const int* dCounts, int arrays, int maxCount);

// Histogram scene-cut statistics of one 4:2:0 picture.
struct HistJob
{
    const void* lowres0;      // origin of the slot's half-res plane 0
    int lowresStride;
    const void* y; const void* u; const void* v;
    long long strideY, strideC;              // pixels
    int width, height, depth;
    uint32_t* hist;                          // [16 segments][3 planes][256] raw sample counts (zeroed)
    unsigned long long* sums;                // [16][3] raw sample sums (zeroed)
    unsigned long long* rowVar;              // [3][maxRows] block-variance sums per block row (zeroed)
    int maxRows;
};
hipError_t launch_hist_stats(hipStream_t s, int bpp, const HistJob& job);
} // namespace xg
struct x265gpu_hist_stats_t;
namespace xg {
// raw counts -> Lowres::picHistogram / averageIntensity* / picAvgVariance* (aq_host.cpp)
void hist_finish(int width, int height, const uint32_t* hist, const unsigned long long* sums,
                 const unsigned long long* rowVar, int maxRows, x265gpu_hist_stats_t* out);
hipError_t launch_cost(hipStream_t s, const Geometry& g, int bpp, const CostJob* dJobs, int n);

// ---- MCSTF motion search: one pyramid level of one picture against numRefs references ----
struct McstfLevel
{
    const void* src;          // the picture's plane of this level, pixel (0,0)
    int stride;               // shared by the picture and its references
    int width, height;        // the level's size as the reference passes it (bounds of the coarser-level lookup)
    int depth;
    int blocksX, blocksY;     // blocks searched (16x16 on levels 0..2, 8x8 on level 3)
    int mvStride;             // blocks per row of the vector arrays
    int prevStride, factor;   // coarser level's vector array stride and scale
};
struct McstfRef
{
    const void* buf;          // the reference picture's plane of this level, pixel (0,0)
    int32_t* mvs;             // {x,y} per block, 1/16 pixel
    const int32_t* prev;      // the coarser level's vectors, or null on level 0
    int32_t* least;           // running best error per block
    int32_t* error;           // level 3: the normalised error that the filter reads
    int32_t* spec;            // level 3: per block { own x, own y, error of left's vector, error of above's } (4 ints)
};
hipError_t launch_mcstf_level(hipStream_t s, int bpp, int level, const McstfLevel& lv, const McstfRef* dRefs, int numRefs);
hipError_t launch_weight_cost(hipStream_t s, const Geometry& g, int bpp, const WeightCostJob* dJob);
hipError_t launch_weight_planes(hipStream_t s, const Geometry& g, int bpp, int depth, int nPlanes, const void* src,
                                void* dst, int w0, int round, int shift, int offset);

} // namespace xg
