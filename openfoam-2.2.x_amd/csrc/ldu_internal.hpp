// Internal structures of libldugpu (not part of the C ABI).
//
// Data layout in HBM:
//   * cells are renumbered into DEPENDENCY-LEVEL order of the lower-triangular DAG
//     (level(c) = 1 + max level of lower neighbours); rows of one level are contiguous,
//     so every triangular sweep (DIC/DILU/GaussSeidel) is a sequence of fully parallel,
//     fully coalesced level kernels that reproduce the reference's sequential result
//     bit-for-bit (per-row accumulation order = the reference's face order);
//   * rows are cut into slices of <= 64 rows (one wavefront) that never straddle a level;
//     a slice stores its off-diagonal entries column-major (entry k of lane i at
//     ent + k*64 + i) = sliced ELL, so each wave-instruction reads 64 consecutive values;
//   * per entry: int32 column (new numbering) + f64 coefficient; entries of a row are in the
//     reference's accumulation order: lower-neighbour faces ascending, then owned faces
//     ascending (lduMatrixATmul.C:75-79).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/ldugpu.h"

#define LDU_WAVE 64
#define LDU_MAX_PEERS 16

void ldu_set_error(const std::string& msg);
#define LDU_CHECK_HIP(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            ldu_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));      \
            return -1;                                                             \
        }                                                                          \
    } while (0)

// hipMemset on device memory is ASYNCHRONOUS with respect to the host and runs on the null stream, which the library's
// non-blocking streams do not wait for: a set-up memset could land AFTER an upload or a sweep enqueued behind it on the
// context's stream (found by tools/fuzz_peer.py in round 4: interfaceIntCoeffs zeroed again after ldu_matrix_set_patch_coeffs,
// one case in ~100 with several processes on one GPU).  Every set-up memset goes through this: set, then drain the null stream.
static inline hipError_t ldu_memset_sync(void* p, int v, size_t n)
{
    hipError_t e = hipMemset(p, v, n);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(nullptr);
}

struct ldu_comm_impl;  // RCCL wrapper (ldu_comm.cpp)
struct ClusterPlan;    // ldu_cluster.hip
struct ClGreedy;       // ldu_cluster_greedy.hpp
struct BlockPlan;      // ldu_blocks.hip
#define LDU_CL_MAXD 12  // ldu_cluster.hip: dependencies per row the cluster kernels hold in registers
#define LDU_PROF_NCAT 8

// scalar slots on the device
enum {
    S_WARA0 = 0, S_WARA1 = 1, S_WAPA = 2, S_RES = 3, S_NORM = 4, S_SINGULAR = 5, S_SUMPSI = 6,
    S_TMP0 = 7, S_TMP1 = 8, S_SCALE_NUM = 9, S_SCALE_DEN = 10, S_COUNT = 11,
    S_STOP = 12,   // Krylov loops: 1 = the iteration whose residual was just formed is the last one (decided on the device)
    S_INIT = 13,   // ... their initial residual (for the relTol test on the device)
    S_BANK = 16, S_NSLOTS = 128
};

struct ldu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // second stream (PBiCG transpose system)
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    // third stream: the coefficient agglomeration of a GAMG hierarchy (levels 1 ...) runs beside the finest level's own work
    // of a solve (its level layout, the initial residual, the restrictions); evAggFork = the finest LDU arrays are in place
    // (recorded by ldu_matrix_set_coeffs for matrix aggForkOf at epoch aggForkEpoch), evAggJoin = the level matrices are ready
    hipStream_t stream3 = nullptr;
    hipEvent_t evAggFork = nullptr, evAggJoin = nullptr;
    const void* aggForkOf = nullptr;
    uint64_t aggForkEpoch = 0;
    int aggOverlap = 1;              // LDU_AGG_OVERLAP=0: everything on the main stream
    int aggPrefill = 1;              // LDU_AGG_PREFILL=0: the engines' coefficient copies are filled by the first smoothing call
    // halo exchange overlapped with the interior rows: the send/recv of an operator application runs on its own
    // stream between pack (initMatrixInterfaces) and apply (updateMatrixInterfaces) - the window the reference
    // itself leaves for the interior loops (lduMatrixUpdateMatrixInterfaces.C:30-93, 127-160; lduMatrixATmul.C:62-89)
    hipStream_t streamComm = nullptr;
    hipEvent_t evPacked = nullptr, evHalo = nullptr;
    bool haloInFlight = false;       // an exchange was started and nobody waited for it yet
    int haloOverlap = 1;             // LDU_HALO_OVERLAP=0: exchange on the compute stream (round-1 behaviour)
    long nHaloOverlapped = 0;        // exchanges that ran on the comm stream (test / report introspection)
    double* d_partials = nullptr;    // reduction scratch [2 * maxBlocks]
    int maxRedBlocks = 1024;
    double* d_scalars = nullptr;     // [S_NSLOTS] = banks of S_BANK slots
    double* h_scalars = nullptr;     // pinned mirror
    // PCG / PBiCG with the convergence read-back off the critical path (ldu_solvers.cpp: solve_krylov): two more pinned
    // mirrors and the events behind their copies - iteration k + 1 is queued before the residual of iteration k is read
    double* h_ring[2] = {nullptr, nullptr};
    hipEvent_t evRing[2] = {nullptr, nullptr};
    int krylovSpeculate = 1;         // LDU_KRYLOV_SPECULATE=0: read the residual before queueing the next iteration (round 1-4)
    int sb = 0;                      // current bank offset (nested solves push a new bank)
    double* S() const { return d_scalars + sb; }
    bool useGraphs = true;
    int fuseRows = 4096;             // levels up to this many rows are fused into one-block chains
    // sweep engine: 1 = persistent point-to-point kernel (default), 0 = one kernel per level
    int sweepP2P = 1;
    int p2pGate = 0;                 // slice-completion gate before granule polling (measured slower: off)
    int p2pTrace = 0;                // diagnostic kernels with per-slice tracing
    int dualStream = 1;              // PBiCG: A system and transposed system on two streams
    int dualActive = 0;              // two cluster sweeps share the chip right now: two workgroups per CU each
    int gsPipeline = 1;
    int gsPipelineMaxSkew = 1 << 30; // (round 1: 4) with the run-ahead window pipelined sweeps no longer starve at large skew; LDU_GS_MAXSKEW              // pipeline consecutive GaussSeidel sweeps in one launch
    int p2pBlocksPerCU = 2;          // measured best on MI355X (fewer pollers): tools/sweep_probe.py
    double p2pWindowLevels = 8.0;    // run-ahead window of the level engines in dependency levels (LDU_P2P_WINDOW; 0 = off)
    // cluster (row-blocking) sweep engine, ldu_cluster.hip
    int clusterEngine = 1;           // LDU_CLUSTER=0: off
    int clusterMinCells = 15000;     // LDU_CLUSTER_MIN
    double clusterPaysFactor = 1.75; // LDU_CLUSTER_PAYS: pipelined sweeps go to the cluster engine when it has this many times fewer levels
    int clusterBlocksPerCU = 3;      // LDU_CLUSTER_BPC (216^3 DIC half sweep: 0.494 / 0.467 / 0.487 / 0.526 ms at 2 / 3 / 4 / 6)
    int clusterBlocksPerCUMulti = 3; // LDU_CLUSTER_BPC_MULTI (pipelined sweeps; 216^3 bench with one ticket counter: 106 / 118.4 / 119.8 / 119.1 V-cycles/s at 1 / 2 / 3 / 4; with eight: 133.2 / 132.6 at 3 / 4)
    int clusterBpcForced = 0;
    int clusterMulti = 1;            // pipelined GaussSeidel sweeps on the cluster engine (LDU_CLUSTER_MULTI=0: off)
    unsigned long long valStamp = 1; // bumped whenever a SELL value array is rewritten
    int smallKernels = 1;            // single-wavefront LDS kernel for tiny matrices (LDU_SMALL=0: off)
    long nHaloExchanges = 0, nAllReduces = 0, nScalarReadbacks = 0;   // communication counters (ldu_ctx_comm_counters)
    int lagBucketWidth = 8;          // LDU_LAG_BUCKETS: width (levels) of the lag buckets rows are grouped by; 0 = off
    int coopRows = 1;                // LDU_COOP_ROWS: several lanes per row for rows with more than 8 lower / upper neighbours
    int sortRowsByWidth = 1;         // LDU_SORT_ROWS: rows of a level ordered by width class (narrow slices stay narrow)
    int smallMaxCells = 6000;        // LDU_SMALL_MAX (<= 16384); single sweeps: the one-wavefront kernel up to 3000 cells
    int gsWideUpper = 1;             // LDU_GS_WIDE_UPPER=0: upper parts of more than 8 entries after the lower part (round-1 order)
    int stageOverlap = 1;            // LDU_STAGE_OVERLAP=0: host vectors of a solve uploaded on the main stream, before any device work
    int clusterDirectFill = 1;       // LDU_CLUSTER_DIRECT=0: cluster layout via the level layout (cl_convert) instead of from the faces
    int wgEngine = 1;                // LDU_WG=0: no one-workgroup engine (solution vector in LDS, k sweeps as LDS-synchronised tasks)
    int wgMaxCells = 6000;           // LDU_WG_MAX (<= 18 000: 9 bytes of LDS per cell; above ~6000 cells one CU is too little)
    int wgMinCells = 0;              // LDU_WG_MIN
    int wgWaves = 8;                 // LDU_WG_WAVES (4 / 8)
    int wgWide = 0;                  // LDU_WG_WIDE=1: also levels with rows wider than 16 entries
    // block engine (ldu_blocks.hip): k pipelined GaussSeidel sweeps with the hand-offs inside a workgroup's LDS, blocks of a few
    // thousand cells, granules only between blocks
    int blkEngine = 1;               // LDU_BLK=0: off
    int blkMinCells = 100;           // LDU_BLK_MIN (below: the one-workgroup engine)
    int blkWideFrom = 400000;        // LDU_BLK_WIDE_FROM: seven compute wavefronts per block from this many cells, three below
    int blkEqualMax = 2600000;       // above: blobs of equal footprint right away (LDU_BLK_EQUAL_MAX), see bk_build
    int blkBySlots = 1;              // LDU_BLK_BY_SLOTS=0: no footprint-balanced blobs (levels that do not fit stay on the level engines)
    int blkMaxCells = 4000000;       // LDU_BLK_MAX (above: more blocks than resident workgroups, and on the motorBike levels nearly all of
                                     // them are open at once - 348 of 349 at 3.1 M cells, 678 of 702 at 6.3 M: profiles/r05_block_open_counts.log)
    int blkCells = 0;                // LDU_BLK_CELLS: cells per block (0 = sized so that all blocks are resident at once)
    int blkCellsMin = 1024;          // LDU_BLK_CELLS_MIN
    int blkCellsMax = 9000;          // LDU_BLK_CELLS_MAX (LDS: 9 bytes per local row and per ghost)
    int blkWaves = 0;                // LDU_BLK_WAVES (7 / 3 compute wavefronts per block, + 1 importer; 0 = by size)
    int deviceSharers = 1;           // contexts that run on this GPU at the same time (peer ranks with the same PCI address; LDU_DEVICE_SHARERS)
    int blkMaxPerCU = 4;             // LDU_BLK_PER_CU: workgroups per CU the grid may count on
    int blkWavesPerSweep = 0;        // LDU_BLK_WPS: wavefronts a sweep's tasks of one block are dealt to (0 = all tasks round-robin over all wavefronts)
    int blkXcdMap = 1;               // LDU_BLK_XCD=0: blocks in launch order instead of contiguous ranges per XCD
    bool gsLayouts = false;          // LDU_GS_LAYOUTS=1: per-sweep layouts for the chip-wide pipelined GaussSeidel sweeps (ldu_gslayouts.cpp);
                                     // off: measured slower on the levels that would take them (see the file's header)
    int gsLayoutsMinCells = 200000;  // LDU_GS_LAYOUTS_MIN
    int blkLayouts = 4;              // LDU_BLK_LAYOUTS: own grouping (by the row's time in the DAG of the k sweeps) for the first n sweeps
    int smallPipe = 1;               // LDU_SMALL_PIPE=0: k sweeps one after the other in ONE wavefront (round-1 kernel)
    int p2pBpcForced = 0;            // LDU_P2P_BPC given: the slab engine does not size its own grid
    int numCUs = 256;
    int p2pMaxBlocksPerCU = 5;       // register-limited residency of the sweep kernels
    double gsmWideSlices = 800.0;    // LDU_GSM_WIDE: slices per dependency level x sweeps in flight from which the pipelined GaussSeidel kernel runs three workgroups per CU instead of two
    // XCD-slab sweep engine: -1 = choose per addressing, 0 = chip-wide engine only, 1..8 = forced
    int p2pSlabs = -1;
    int nXcd = 0;                    // XCDs seen by the placement census (0 = census failed: no slabs)
    int* d_abort = nullptr;          // set by a sweep whose bounded spin expired
    int* h_abort = nullptr;          // pinned mirror
    int p2pGen = 0;                  // bumped when a sweep aborted: addressings reset their tickets
    int abortSeen = 0;               // an aborted sweep was detected since run_with_fallback() cleared this
    long nFallbacks = 0;             // operations re-run on the level-kernel engine after an aborted sweep
    long nDiscardedAborts = 0;       // sweeps of a discarded (speculative) Krylov iteration that gave up waiting: dropped, not an error
    int consecFallbacks = 0;         // ... in a row (three: the context stays on the level kernels)
    // communicator
    int rank = 0, nRanks = 1;
    ldu_comm_impl* comm = nullptr;
    int commEpoch = 0;               // bumped when the carriers of halos / sums change (ldu_ctx_comm_select)
    // per-kernel-class timing with HIP events on the compute stream (bench.py roofline)
    bool profOn = false;
    const struct ldu_addr* profAddr = nullptr;
    struct ProfCat { std::vector<hipEvent_t> ev; size_t used = 0; long launches = 0; };
    ProfCat prof[LDU_PROF_NCAT];
    void profStart(const struct ldu_addr* a, int cat);
    void profStop(const struct ldu_addr* a, int cat);
};

struct Segment {
    int levelBegin, levelEnd;   // [begin, end) levels
    int sliceBegin, sliceEnd;   // [begin, end) slices
    bool fused;                 // one block walks the levels with __syncthreads()
};

struct Patch {
    int n = 0;
    int nbrRank = -1;
    int nbrPatch = -1;            // >= 0: cyclic - the neighbour is patch nbrPatch of the SAME addressing
    std::vector<int> faceCells;   // original numbering
    int* d_faceCells = nullptr;   // new numbering
    double* d_send = nullptr;
    double* d_recv = nullptr;
    int offset = 0;               // offset of this patch in the concatenated patch-face arrays
};

// peer-store halo of an addressing (communication backend "peer", ldu_comm.cpp / ldu_peer.hip): the receive granules of
// every processor patch live in THIS rank's window ([patch][parity][n] x 16 bytes); the neighbour's pack kernel writes
// them directly (xGMI peer stores), this rank's unpack kernel polls them.  Tables are per patch face.
struct PeerHalo {
    size_t winOff = 0, winBytes = 0;       // my receive region in my window
    uint4** d_dst = nullptr;               // [2][nPatchFaces] destination granule per face and parity (null: cyclic face)
    const uint4** d_src = nullptr;         // [2][nPatchFaces] my receive granule per face and parity (null: cyclic face)
    unsigned seq = 0;                      // exchanges started on this addressing (tag of the granules; parity = seq & 1)
    bool pending = false;                  // pack ran, unpack did not yet
    // a second, kernel-private set for addressings small enough to be a coarsest GAMG level: the whole distributed
    // Krylov solve of that level runs in ONE kernel per rank (coarsest_krylov_peer_kernel) that exchanges through these
    // regions with its own sequence number kept on the device
    size_t kWinOff = 0, kWinBytes = 0;
    uint4** d_kdst = nullptr;              // [2][nPatchFaces]; entries stay null when the neighbour has no such region
    const uint4** d_ksrc = nullptr;
    unsigned* d_kseq = nullptr;            // device: exchanges the kernel has done so far
    bool kAll = false;                     // every remote face has a kernel-private destination
    // a third set for the BLOCK engine (ldu_blocks.hip, "Remote interfaces"): the interface values of the k pipelined sweeps of
    // one launch travel through these regions, tagged bSeq + 1 ... bSeq + k (host-side count: every rank launches the same
    // smoothing calls on a level), parity = tag & 1
    size_t bWinOff = 0, bWinBytes = 0;
    uint4** d_bdst = nullptr;              // [2][nPatchFaces]; null entries: cyclic face / the neighbour has no such region
    const uint4** d_bsrc = nullptr;
    unsigned bSeq = 0;
    bool bAll = false;                     // every remote face has a block-engine destination
};
#define LDU_COARSEST_MAXC 64
#define LDU_COARSEST_MAXF 512
#define LDU_COARSEST_MAXP 256
#define LDU_PEERK_MAXCELLS 18000       // kernel-private window regions for addressings up to this size (one-workgroup engine)

struct ldu_addr {
    ldu_ctx* ctx = nullptr;
    int nCells = 0, nFaces = 0;
    // Sweep plans of a GAMG level (block-engine layouts, pipelined task orders: 1.4 of the 3.9 s of the first solve on the
    // 12.7 M-cell motorBike mesh) are built on a host thread BEHIND the first solves (ldu_gamg.cpp: ensure_hierarchy).  While
    // bgState is 1 every engine that would use or build them stands back - the level engines sweep one sweep per launch, bit for
    // bit the same results - and nothing else of this addressing is written by the thread.  2 = built.
    std::thread bgPlan;
    std::atomic<int> bgState{0};
    // host addressing (original numbering)
    std::vector<int> l, u, losort, ownerStart, losortStart;
    std::vector<int> perm, iperm;          // perm[new] = old ; iperm[old] = new
    std::vector<int> level;                // per old cell
    int nLevels = 0;
    std::vector<int> levelStart;           // rows (new numbering), size nLevels+1
    std::vector<int> levelSliceStart;      // size nLevels+1
    int nSlices = 0;
    long nEntries = 0;                     // padded entry count
    int maxRowWidth = 0;                   // max over rows of nL + nU
    std::vector<Segment> segs;
    std::vector<double> faceWeights;
    // sub-domain mode (ldu_addr_set_subdomains): sub-domain of every cell, empty = none.  The cells of a sub-domain are coupled to
    // the others through (cyclic) interfaces only - K ranks of the reference inside one addressing; GAMG then stops coarsening
    // as the K-rank run does (every sub-domain keeps nCellsInCoarsestLevel cells, GAMGAgglomeration.C:53-62)
    std::vector<int> subOf;
    int nSub = 0;
    bool finalized = false;

    // device
    int* d_perm = nullptr;
    int* d_iperm = nullptr;
    int* d_sliceRow = nullptr;             // [nSlices+1] first row of slice (rows of slice s = [r[s], r[s]+cnt))
    int* d_sliceCnt = nullptr;             // [nSlices] rows in slice (<= 64)
    int* d_sliceEnt = nullptr;             // [nSlices] first entry
    int* d_sliceW = nullptr;               // [nSlices] width (max nL+nU)
    bool lagBuckets = false;               // rows of a level also grouped by their pipelined-sweep lag (ldu_plan.cpp)
    std::vector<unsigned short> rowKey;    // [nCells] level row -> (width class, lag bucket) key
    unsigned char* d_sliceT = nullptr;     // [nSlices] lanes per row: 1, or 2 / 4 / 8 in a cooperative slice (ldu_plan.cpp)
    int nCoopSlices = 0;
    int* d_levelSliceStart = nullptr;      // [nLevels+1]
    unsigned char* d_nL = nullptr;         // [nCells] lower-part entries per row (new numbering)
    unsigned char* d_nU = nullptr;         // [nCells]
    int* d_col = nullptr;                  // [nEntries]
    int* d_face = nullptr;                 // [nEntries] original face index (-1 padding)
    int* d_l = nullptr;                    // [nFaces] original lowerAddr
    int* d_u = nullptr;                    // [nFaces]
    int* d_losort = nullptr;               // [nFaces]
    int* d_ownerStart = nullptr;           // [nCells+1]
    int* d_losortStart = nullptr;          // [nCells+1]

    // coupled patches
    std::vector<Patch> patches;
    int nPatchFaces = 0;
    // boundary rows: for every cell touched by a coupled patch, its (patch-face) list in
    // the reference's update order (patch ascending, face ascending)
    int nBRows = 0;
    int* d_bRow = nullptr;                 // [nBRows] row (new numbering)
    int* d_bStart = nullptr;               // [nBRows+1]
    int* d_bFace = nullptr;                // [nPatchFaces] index into concatenated patch-face arrays
    int* d_pfCell = nullptr;               // [nPatchFaces] faceCells (new numbering), concatenated
    // nonBlockingGaussSeidel: per row, index into bRow (-1: no coupled face) and the number of its
    // lower-part entries whose ORIGINAL cell index is below blockStart (= smallest coupled faceCell)
    int* d_nbRowB = nullptr;               // [nCells]
    unsigned char* d_nbK0 = nullptr;       // [nCells]
    double* d_sendAll = nullptr;           // [nPatchFaces]
    double* d_recvAll = nullptr;           // [nPatchFaces]
    PeerHalo* peer = nullptr;              // peer-store backend only
    // small patched level: the k sweeps of a smoothing and their exchanges in ONE launch (gs_wg_peer_kernel); the decision is
    // collective (and-reduce over the ranks) and belongs to a carrier epoch of the context
    int peerWg = -1, peerWgEpoch = -1;
    int peerBlk = -1, peerBlkEpoch = -1;   // block engine with REMOTE interfaces on this addressing: -1 not decided (a collective decision), 0 / 1
    int* d_cycPair = nullptr;              // [nPatchFaces] paired face of a cyclic face, -1 = remote (lazy)

    // point-to-point sweep state: one 16-byte {value lo, tag, value hi, tag} granule per row,
    // a chunk ticket counter and the launch epoch (= tag; never 0)
    // two independent lanes so that two sweeps of the same addressing can run concurrently on
    // two streams (PBiCG: the A system and the transposed system)
    struct P2PLane {
        uint4* d_X = nullptr;              // [nCells] write-through copies of exported rows (slab engine)
        unsigned* d_ctl = nullptr;         // [2][8] per-slab chunk tickets, double-buffered by launch parity
        unsigned par = 0;
        uint4* d_granule = nullptr;        // [nCells]
        unsigned* d_ticket = nullptr;      // [64]: [0] tickets, [32] chunks reported complete
        unsigned ticketBase = 0;
        unsigned doneBase = 0;             // advanced by the chunks of every launch that reports (window > 0)
        unsigned epoch = 0;
        int gen = 0;
    };
    P2PLane p2p[2];
    P2PLane* lane(int i);                  // lazily allocates lane 1
    int* d_gateF = nullptr;                // [nSlices] gate slice of forward sweeps (-1 none)
    int* d_gateB = nullptr;                // [nSlices] gate slice of backward sweeps
    unsigned* d_sliceDone = nullptr;       // [nSlices] completion tags (hint for the gate)

    // XCD slabs: contiguous ranges of the ORIGINAL cell numbering (dependencies only run from lower to
    // higher slabs), one per XCD; slices never straddle a slab; slabList = each slab's slices in
    // level order; colX = col with bit 31 set where the column lives in another slab
    int nSlabs = 0;                        // 0 = chip-wide engine
    double slabWidth = 0;                  // average slices per level per slab (sizes the grid)
    int slabStart[9] = {0};                // offsets into slabList
    int slabLevelSpan[8] = {1, 1, 1, 1, 1, 1, 1, 1};   // dependency levels a slab's slices span
    int maxUpper = 0;                      // most upper neighbours of a row
    int* d_slabList = nullptr;             // [nSlices]
    int* d_colX = nullptr;                 // [nEntries]
    unsigned char* d_xflag = nullptr;      // [nCells] 1 = has a neighbour in another slab

    ClusterPlan* cluster = nullptr;        // secondary structure of the cluster sweep engine (lazy)
    BlockPlan* blocks = nullptr;           // secondary structure of the block engine (lazy, ldu_blocks.hip)
    // the greedy clustering of a large addressing starts on a host thread of its own as soon as plan_build knows the
    // dependency levels (it needs nothing else) and runs beside the rest of the level plan; cluster_build joins it
    ClGreedy* greedyEarly = nullptr;
    std::thread greedyThread;
    // single-workgroup pipelined GaussSeidel sweeps of small matrices (gs_small_pipe_kernel): per slice, how many
    // slices the PREVIOUS sweep must have finished before this slice may run (all its upper neighbours done)
    int* d_smallNeed = nullptr;            // [nSlices] (lazy)
    double smallLag = 0;                   // average need[s] - s: how far a sweep trails the previous one

    // topological (sweep, slice) task lists of k pipelined GaussSeidel sweeps, per k
    struct GsTasks { int* d_tasks = nullptr; int n = 0; int* d_slabTasks = nullptr; int slabStart[9] = {0}; bool layouts = false; };
    // Per-sweep layouts of the chip-wide pipelined GaussSeidel sweeps (ldu_gslayouts.cpp): sweep j >= 1 of a launch has its OWN
    // slices - rows grouped by their time T_j in the row-level DAG of the k sweeps (T_0 = dependency level, T_j(r) = 1 + max(T_j of
    // the lower neighbours, T_j-1 of the upper neighbours, T_j-1(r))) - with its own entry tables; rows are addressed through
    // rowIdx (slot -> row of the level numbering), granules / psi / rhs / diag stay in the level numbering.
    struct GsLayout {
        int nSlices = 0; long nEntries = 0;
        int* d_sliceRow = nullptr; int* d_sliceCnt = nullptr; int* d_sliceEnt = nullptr; int* d_sliceW = nullptr;
        unsigned char* d_sliceT = nullptr; unsigned char* d_nL = nullptr; unsigned char* d_nU = nullptr;
        int* d_col = nullptr; int* d_face = nullptr; int* d_rowIdx = nullptr;
        std::vector<int> sliceTime;          // host: T_j of the slice's rows
        bool coop = false;
    };
    GsLayout* gsLay[4] = {nullptr, nullptr, nullptr, nullptr};     // [j], j = 1 .. 3 ([0] = the level layout itself)
    int gsLayState = 0;                    // 0 not decided, 1 built (up to gsLayBuilt sweeps), -1 not on this addressing
    int gsLayBuilt = 0;
    struct GsLayVals { double* d[4] = {nullptr, nullptr, nullptr, nullptr}; unsigned long long stamp[4] = {0, 0, 0, 0}; };
    std::map<const double*, GsLayVals> gsLayVals;     // level value array -> the layouts' value arrays
    // level-layout coefficient arrays filled from face-ordered ones (fill_sell): value array -> (lower-side, upper-side source)
    std::map<const double*, std::pair<const double*, const double*>> valOrigin;
    // when a level-layout value array of THIS addressing was last rewritten (a value of ctx->valStamp): what the engines' converted
    // copies are compared with.  Per array, not per context: the coefficient chain of a GAMG solve fills the level arrays one after
    // the other and the engine copies of a level right behind its array (ensure_hierarchy), on a stream beside the solve
    std::map<const double*, unsigned long long> arrStamp;
    std::map<int, GsTasks> gsTasks;
    struct WgTasks { int* d_tasks = nullptr; int n = 0; int steps = 0; };   // one-workgroup engine: 4 ints per (sweep, slice) task
    std::map<int, WgTasks> wgTasks;
    bool wgLevel = false;                  // small enough for the one-workgroup engine: plain full slices (no lag buckets, no cooperative rows)

    // cached graphs of level-scheduled sweeps, keyed by (mode, pointer arguments)
    std::map<std::string, hipGraphExec_t> graphs;

    // scratch vectors (new numbering), reused by the C-ABI vector entry points
    std::vector<double*> scratch;
    double* scratchVec(int i);
};

struct GamgHierarchy;  // ldu_gamg.cpp

// the patches of an fvMesh for the fvMatrix glue and the fv schemes (ldu_fvmatrix.hip, ldu_fvschemes.hip)
struct ldu_fv_boundary {
    ldu_addr* a = nullptr;
    int nPatches = 0;
    int nFacesTotal = 0;
    std::vector<int> sizes, offsets, coupled;
    int* d_cellStart = nullptr;        // [nCells+1] CSR over cells
    int* d_cellFace = nullptr;         // [nFacesTotal] index into the concatenated patch-face arrays
    int* d_faceCells = nullptr;        // [nFacesTotal]
    unsigned char* d_coupled = nullptr;  // [nFacesTotal] per patch face: its patch is coupled
};


struct ldu_matrix {
    ldu_addr* a = nullptr;
    bool sym = true;
    bool haveCoeffs = false;
    // LDU-space coefficients, original order (device)
    double* d_diagO = nullptr;
    double* d_upperO = nullptr;
    double* d_lowerO = nullptr;     // == d_upperO when symmetric
    bool ownsLdu = true;            // coarse GAMG levels own theirs as well
    // compute layout
    double* d_diag = nullptr;       // [nCells] new numbering
    double* d_valA = nullptr;       // [nEntries] lower part = lower[f], upper part = upper[f]
    double* d_valT = nullptr;       // transpose coefficients (== d_valA when symmetric)
    // coupled patch coefficients, concatenated in patch order
    double* d_bou = nullptr;        // [nPatchFaces]
    double* d_int = nullptr;
    // preconditioner / smoother factor cache (valid for the current coefficients)
    int rDKind = -1;                // LDU_PRE_DIC / LDU_PRE_DILU
    double* d_rD = nullptr;
    double* d_valP = nullptr;       // rD[row]*valA
    double* d_valPT = nullptr;      // rD[row]*valT
    double* d_rDiag = nullptr;      // 1/diag (diagonal preconditioner)
    bool rDiagValid = false;
    // GAMG hierarchy (addressing part cached when cacheAgglomeration)
    GamgHierarchy* gamg = nullptr;
    // work vectors
    std::vector<double*> work;
    double* workVec(int i);
    uint64_t coeffEpoch = 0;
    // LduMatrix<Type,scalar,scalar> solvers on these coefficients (ldu_coupled.hip)
    struct CoupledWork* coupled = nullptr;
    // directSolveCoarsest with coupled patches / several ranks: the gathered coarsest level (ldu_coarsest.hip)
    struct CoarsestLU* lu = nullptr;
};
void coupled_free(ldu_matrix* m);
void coarsest_lu_free(ldu_matrix* m);
int dev_halo_start(ldu_matrix* m, const double* x);   // initMatrixInterfaces: pack + exchange

// ---------------------------------------------------------------- kernels (ldu_kernels.hip)
enum SweepMode {
    SW_TRI_FWD = 0,   // w = rD*rhs - sum_lower valP*w[col]            (DIC/DILU forward)
    SW_TRI_BWD = 1,   // w -= sum_upper(desc) valP*w[col]              (DIC/DILU backward)
    SW_RD = 2,        // rD = diag - sum_lower (valT*valA)/rD[col]     (calcReciprocalD, unreciprocated)
    SW_GS_FWD = 3,    // GaussSeidel forward (optionally stores bPrime)
    SW_GS_BWD = 4,    // symGaussSeidel reverse sweep from the stored bPrime
    // The templated LduMatrix<Type,scalar,scalar> family associates differently (TDILUPreconditioner.C:67-70,
    // :111-120; TGaussSeidelSmoother.C:139): the row factor multiplies the finished product, so the
    // coefficients cannot be pre-scaled.  val = the unscaled coefficients, scale = rD.
    SW_TRI_FWD_T = 5, // w = rD*rhs - sum_lower rD*(val*w[col])
    SW_TRI_BWD_T = 6, // w -= sum_upper(desc) rD*(val*w[col])
    SW_RD_T = 7,      // rD = diag - sum_lower (valT*valA)*(1/rD[col])
    SW_GS_FWD_T = 8   // GaussSeidel forward, finished with rD*acc instead of acc/diag
};
// mode without the association flavour / is it the templated flavour
constexpr int sw_base(int m) { return m == 5 ? 0 : m == 6 ? 1 : m == 7 ? 2 : m == 8 ? 3 : m; }
constexpr bool sw_tform(int m) { return m >= 5; }

struct SweepArgs {
    int mode;
    double* w;            // output / in-place vector (w, rD or psi)
    const double* rhs;    // rhs (TRI_FWD), source/bPrime (GS_FWD), stored bPrime (GS_BWD)
    const double* scale;  // rD (TRI_FWD) or diag (RD, GS_*)
    const double* val;    // valP / valA
    const double* val2;   // valT for SW_RD
    double* aux;          // bPrime store (GS_FWD when non-null)
    int lane;             // P2P state lane (0 default; 1 = second concurrent sweep)
    hipStream_t stream;   // nullptr = the context's compute stream
};

int k_sweep(ldu_addr* a, const SweepArgs& args);
int k_set_p2p_sleep(int n);
int k_xcd_census(ldu_ctx* ctx);
int k_sweep_cluster(ldu_addr* a, const SweepArgs& g, hipStream_t s);   // 1 = not taken
int k_cluster_prebuild(const std::vector<ldu_addr*>& addrs);
int k_cluster_build_one(ldu_addr* a);                                   // one addressing (own set-up thread)            // cluster plans of several addressings, in parallel
std::string ldu_last_error_string();
int k_sweep_cluster_vec3(ldu_addr* a, int mode, double* w, const double* rhs, size_t stride, const double* scale,
                         const double* val, int lane, hipStream_t s);   // three component planes at once; 1 = not taken
bool k_cluster_active(ldu_addr* a);
bool k_cluster_kind_active(ldu_addr* a, int kind);
int k_engine_of(ldu_addr* a, int kind);
int k_gs_prebuild(ldu_addr* a, int k);   // the host plan of k pipelined GaussSeidel sweeps on the level engines, ahead of the first call
int k_sweep_cluster_gs_multi(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* valA);
void cluster_free(ldu_addr* a);
// block engine (ldu_blocks.hip)
int k_sweep_gs_blocks(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val);   // 1 = not taken
bool k_blocks_active(ldu_addr* a);
bool comm_peer_carries_halo(const ldu_ctx* ctx);
int k_blocks_peer_decide(ldu_addr* a);     // collective: remote interfaces on the block engine (ldu_blocks.hip)
int k_blocks_prebuild(ldu_addr* a, int k);
int k_blocks_prefill(ldu_addr* a, const double* val, const double* bou, hipStream_t s);   // the layouts' coefficient copies, ahead of the sweep (1 = nothing to fill)
int k_cluster_prefill(ldu_addr* a, const double* val, hipStream_t s);
int dev_smooth_prefill(ldu_matrix* m, int smoother, hipStream_t s);
int k_blocks_set_watchdog(unsigned long long budgetTicks, unsigned long long stallTicks);
int k_blocks_set_trace(unsigned long long* buf);
int k_blocks_info(ldu_addr* a, int k, long out[8]);
void blocks_free(ldu_addr* a);
extern "C" int partition_blobs(int nCells, int nFaces, const int* lowerAddr, const int* upperAddr, int nParts, int* part);   // ldu_mesh.hip
extern "C" int partition_blobs_slots(int nCells, int nFaces, const int* lowerAddr, const int* upperAddr, long slotTarget, int maxParts, int* part);
void blocks_forget(ldu_addr* a, const double* levelVal);
void gs_layouts_free(ldu_addr* a);                                    // ldu_gslayouts.cpp
void gs_layouts_forget(ldu_addr* a, const double* levelVal);
void cluster_forget(ldu_addr* a, const double* levelVal);   // drop the converted copy of a value array
int k_sweep_gs_nonblocking(ldu_addr* a, double* psi, const double* source, const double* diag, const double* val,
                           const double* bou);
int k_coarsest_solve(ldu_matrix* A, double tolerance, double relTol, int maxIter, double* corr, const double* src);
int k_coarsest_lu(ldu_matrix* A, double* corr, const double* src, uint64_t epoch);
int k_coarsest_solve_peer(ldu_matrix* A, double tolerance, double relTol, int maxIter, double* corr, const double* src,
                          const int* d_cycPair);   // ldu_coarsest.hip: the distributed solve in one kernel per rank; 1 = not taken
bool k_coarsest_peer_eligible(ldu_matrix* A);   // directSolveCoarsest (ldu_coarsest.hip)
int k_sweep_gs_wg(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val);
int k_sweep_gs_wg_peer(ldu_addr* a, int k, double* psi, const double* source, const double* diag, const double* val,
                       const double* bou, const int* d_cycPair);   // with coupled patches, k sweeps + exchanges in one launch; 1 = not taken
bool k_wg_peer_eligible(ldu_addr* a);
int k_set_peer_timeout_kernels(unsigned long long ticks);
int k_sweep_gs_small(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val);
int k_set_p2p_backoff(unsigned n);
int k_set_p2p_backoff_cap(unsigned n);
int k_read_p2p_dbg(int* out);
int k_read_p2p_dbg_records(int* out);
int k_set_p2p_trace(unsigned long long* buf);
int k_sweep_gs_multi(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag,
                     const double* val);

int k_fill_sell(ldu_addr* a, const double* lowerO, const double* upperO, double* val, hipStream_t s);
int k_permute_in(ldu_addr* a, double* dstNew, const double* srcOld, hipStream_t s);   // dst[new] = src[perm[new]]
int k_permute_out(ldu_addr* a, double* dstOld, const double* srcNew, hipStream_t s);  // dst[perm[new]] = src[new]
int k_scale_rows(ldu_addr* a, double* valOut, const double* valIn, const double* rowScale, hipStream_t s);
int k_reciprocal(int n, double* x, hipStream_t s);
int k_amul(ldu_matrix* m, double* y, const double* x, bool transpose, hipStream_t s);
int k_residual_rows(ldu_matrix* m, double* r, const double* x, const double* b, hipStream_t s);
int k_sumA_rows(ldu_matrix* m, double* sumA, hipStream_t s);
int k_offdiag(ldu_matrix* m, double* y, const double* x, int mode, hipStream_t s);  // H (0), H1 (1), interpolate (2)
int k_faceH(ldu_matrix* m, double* faceH, const double* xOld, hipStream_t s);

// interfaces: pack psi[faceCells] -> send buffers; apply result[row] -= sign*coeff*recv
int k_pack_patches(ldu_addr* a, const double* x, hipStream_t s);
int k_apply_patches(ldu_addr* a, double* result, const double* coeffs, double sign, hipStream_t s);
int k_apply_patches_from(ldu_addr* a, double* out, const double* in, const double* coeffs, double sign, hipStream_t s);
int k_sumA_patches(ldu_addr* a, double* sumA, const double* bou, hipStream_t s);

// elementwise
enum EwOp {
    EW_COPY = 0,        // y = a
    EW_SUB = 1,         // y = a - b
    EW_ADD_INPLACE = 2, // y += a
    EW_MUL_INPLACE = 3, // y *= a
    EW_ZERO = 4,        // y = 0
    EW_DIV = 5,         // y = a / b
    EW_MUL = 6,         // y = a * b
    EW_SUB_INPLACE = 7  // y -= a
};
int k_ew(int n, int op, double* y, const double* a, const double* b, hipStream_t s);
int k_pcg_update_p(int n, double* pA, const double* wA, const double* scalars, int cur, int prev,
                   int first, hipStream_t s);
int k_pbicg_update_p(int n, double* pA, const double* wA, double* pT, const double* wT,
                     const double* scalars, int cur, int prev, int first, hipStream_t s);
// psi += alpha pA ; rA -= alpha wA ; (rT -= alpha wT) ; partial sum |rA| ; singular test on device
int k_pcg_update_xr(ldu_ctx* ctx, int n, double* psi, double* rA, const double* pA, const double* wA,
                    double* rT, const double* wT, int cur, hipStream_t s);
// S_STOP := singular || converged || !(it < maxIter)  (PCG.C:174-181's loop condition, evaluated on the device)
int k_krylov_decide(ldu_ctx* ctx, double tolerance, double relTol, int it, int maxIter, hipStream_t s);
int k_gamg_scale_update(int n, double* field, const double* source, const double* Acf,
                        const double* diag, const double* scalars, hipStream_t s);
int k_neg_div(int n, double* psi, const double* Apsi, const double* diag, hipStream_t s);

// reductions -> ctx->d_scalars[slot] (deterministic two-stage tree)
enum RedOp { RED_DOT = 0, RED_SUMMAG = 1, RED_SUM = 2, RED_NORMFACTOR = 3, RED_DOT2 = 4 };
int k_reduce(ldu_ctx* ctx, int n, int op, const double* a, const double* b, const double* c,
             const double* d, int slot, hipStream_t s);

// GAMG transfer
int k_restrict(int nCoarse, const int* childStart, const int* child, const double* fine,
               double* coarse, hipStream_t s);
int k_prolong(int nFine, const int* map, const double* coarse, double* fine, hipStream_t s);
int k_agglomerate_coeffs(int nCoarseFaces, const int* cfStart, const int* cfFine, const unsigned char* cfFlip,
                         int nCoarseCells, const int* ccStart, const int* ccFine,
                         const int* childStartO, const int* childO,
                         const double* fineDiag, const double* fineUpper, const double* fineLower,
                         double* coarseDiag, double* coarseUpper, double* coarseLower, bool sym,
                         hipStream_t s);

// fv stencils (face loops) on the original numbering
int k_fv_interpolate(ldu_addr* a, int nComp, const double* lambdas, const double* vf, double* sf, hipStream_t s);
int k_fv_surfaceIntegrate(ldu_addr* a, int nComp, const double* ssf, const double* sfVec, const double* V,
                          double* out, hipStream_t s);
int k_fv_snGrad(ldu_addr* a, const double* delta, const double* vf, double* ssf, hipStream_t s);
int k_fv_coeffs_diag(ldu_addr* a, int mode, const double* A, const double* B, double* lower, double* upper, double* diag,
                     hipStream_t s);   // 1 = not applicable
int k_fv_negSumDiag(ldu_addr* a, const double* lower, const double* upper, double* diag, hipStream_t s);
int k_fv_laplacian_coeffs(int nFaces, const double* delta, const double* gammaMagSf, double* upper, hipStream_t s);
int k_fv_div_coeffs(int nFaces, const double* w, const double* phi, double* lower, double* upper, hipStream_t s);

// ---------------------------------------------------------------- host pieces
int plan_build(ldu_addr* a);
// stamp of a value array (an array nobody registered: the context-wide counter - every rewrite anywhere invalidates its copies)
inline unsigned long long val_stamp(const ldu_addr* a, const double* v)
{
    auto it = a->arrStamp.find(v);
    return it == a->arrStamp.end() ? a->ctx->valStamp : it->second;
}
inline void val_touch(ldu_addr* a, const double* v) { a->arrStamp[v] = ++a->ctx->valStamp; }                              // ldu_plan.cpp
int plan_finalize_patches(ldu_addr* a);
void plan_free(ldu_addr* a);
extern thread_local bool tl_bgPlanThread;      // set on a background plan thread: it IS the builder, the guards below let it through
inline bool addr_bg_pending(const ldu_addr* a) { return !tl_bgPlanThread && a->bgState.load(std::memory_order_acquire) == 1; }
inline void addr_bg_wait(ldu_addr* a) { if (a->bgPlan.joinable()) a->bgPlan.join(); }

int comm_allreduce_scalars(ldu_ctx* ctx, int slot, int count, hipStream_t s);   // ldu_comm.cpp
int comm_exchange(ldu_addr* a, hipStream_t s);                                   // halo send/recv (may return before it ran)
int comm_wait_halo(ldu_addr* a, hipStream_t s);                                  // s waits for the exchange started last on a
bool comm_is_peer(const ldu_ctx* ctx);                                            // peer-store backend active for halos
int comm_peer_setup_addr(ldu_addr* a);                                            // plan time: receive region + tables
void comm_peer_free_addr(ldu_addr* a);
int comm_halo_pack_exchange(ldu_addr* a, const double* x, hipStream_t s);         // initMatrixInterfaces: pack + start the exchange
// ldu_peer.hip
int k_peer_pack(ldu_addr* a, const double* x, unsigned seq, hipStream_t s);
int k_peer_unpack(ldu_addr* a, unsigned seq, hipStream_t s);
int k_peer_set_timeout(double seconds);
int k_coarsest_set_peer_timeout(unsigned long long ticks);   // ldu_coarsest.hip
struct PeerRed { uint4* win[LDU_MAX_PEERS]; };
int k_peer_allreduce(ldu_ctx* ctx, const PeerRed& P, size_t redOff, int me, int n, int count, unsigned seq, double* vals,
                     int* abortWord, hipStream_t s);
// everything the in-kernel collectives of a context need (ldu_comm.cpp fills it)
struct PeerKernelComm { PeerRed P; size_t redOff; unsigned* d_redSeq; int me, n; };
bool comm_peer_kernel_comm(ldu_ctx* ctx, PeerKernelComm* out);   // false: peer stores do not carry halos AND sums
int comm_allreduce_min_int(ldu_ctx* ctx, int* v);
int comm_allgather_host(ldu_ctx* ctx, const void* mine, int64_t nBytes, std::vector<std::vector<char>>& all);
int comm_allreduce_abort(ldu_ctx* ctx, hipStream_t s);                             // abort flag := max over the ranks
int comm_exchange_ints(ldu_ctx* ctx, const std::vector<Patch>& patches,
                       const std::vector<std::vector<int>>& send, std::vector<std::vector<int>>& recv);
int k_patch_agglomerate(int nCoarse, const int* start, const int* fine, const double* fBou,
                        const double* fInt, double* cBou, double* cInt, hipStream_t s);
void comm_destroy(ldu_ctx* ctx);

// solvers (ldu_solvers.cpp): all vectors device, new numbering
int matrix_ensure_rD(ldu_matrix* m, int kind);
int dev_amul(ldu_matrix* m, double* y, const double* x, bool transpose, hipStream_t s = nullptr);
int dev_residual(ldu_matrix* m, double* r, const double* x, const double* b);
int dev_sumA(ldu_matrix* m, double* sumA);
int dev_precondition(ldu_matrix* m, int kind, double* w, const double* r, bool transpose, hipStream_t s);
int dev_smooth(ldu_matrix* m, int smoother, double* psi, const double* source, int nSweeps);
int dev_solve(ldu_matrix* m, const ldu_controls* c, double* psi, const double* source, ldu_perf* perf,
              double* hist);
int dev_read_scalars(ldu_ctx* ctx, int slot, int count, double* out);
int dev_check_abort(ldu_ctx* ctx);
// Engine fallback.  The point-to-point / cluster engines bound every dependency wait; a sweep whose bound
// expires drains and flags the context (the reference's sweeps simply always complete,
// DICPreconditioner.C:87-123, GaussSeidelSmoother.C:66-187).  An operation that saw such an abort is run
// again from its (untouched) inputs on the level-kernel engine, whose results are bit-identical.
// Multi-rank: the flag is max-reduced over the ranks before it is read (comm_allreduce_abort), so the re-run is
// taken by every rank or by none and the sequence of collectives stays in step.
int fallback_prepare(ldu_matrix* m);           // drain, clear the flag, drop sweep-built caches; 0 = ok
template <class F>
static inline int run_with_fallback(ldu_matrix* m, F&& op)
{
    ldu_ctx* ctx = m->a->ctx;
    ctx->abortSeen = 0;
    int rc = op();
    if (!ctx->abortSeen) { ctx->consecFallbacks = 0; return rc; }
    if (fallback_prepare(m)) return -1;
    const int p2p = ctx->sweepP2P;
    ctx->sweepP2P = 0;
    rc = op();
    // a context whose fast engines gave up three operations in a row (a GPU shared with other processes, a profiler, a
    // debugger: healthy launches that exceed the wait budget) stays on the level kernels instead of paying the failed
    // attempt, the cache drop and the re-run on every operation
    if (++ctx->consecFallbacks >= 3 && p2p)
    {
        fprintf(stderr, "[ldugpu] warning: three consecutive operations exceeded the dependency-wait budget; this context "
                        "keeps the level-kernel engine from now on (LDU_WATCHDOG_MS / ldu_ctx_set_watchdog raise the budget)\n");
        return rc;
    }
    ctx->sweepP2P = p2p;
    return rc;
}
int k_div_check(ldu_ctx* ctx, unsigned long long seed, long n, unsigned long long* mismatches);
int k_stream(ldu_ctx* ctx, int mode, long n, int reps, double* seconds);
int k_set_p2p_wide(int on);
int k_set_gs_multi_trace(unsigned long long* buf, int nSlices);
int k_set_watchdog(unsigned long long budgetTicks, unsigned long long stallTicks);           // ldu_kernels.hip
int k_cluster_set_watchdog(unsigned long long budgetTicks, unsigned long long stallTicks);   // ldu_cluster.hip
int k_set_spin_limit(unsigned polls);          // ldu_kernels.hip (0 = default)
int k_cluster_set_spin_limit(unsigned polls);  // ldu_cluster.hip
int k_cluster_set_trace(unsigned long long* buf);
int k_cluster_levels(ldu_addr* a, int* out, int cap);
void gamg_invalidate_factors(GamgHierarchy* g);
void coupled_invalidate(ldu_matrix* m);

// GAMG (ldu_gamg.cpp)
int gamg_solve(ldu_matrix* m, const ldu_controls* c, double* psi, const double* source, ldu_perf* perf,
               double* hist);
int gamg_precondition_setup(ldu_matrix* m, const ldu_controls* c);
int gamg_precondition(ldu_matrix* m, const ldu_controls* c, double* wA, const double* rA);
void gamg_free(GamgHierarchy* g);
int gamg_build_for_query(ldu_matrix* m, const ldu_controls* c);
int gamg_query(ldu_matrix* m, int32_t* nLevels, int32_t* nCells, int32_t* nFaces);
int gamg_level_info(ldu_matrix* m, int level, int32_t out[8]);
void gamg_wait_plans(GamgHierarchy* g);
int gamg_level_data(ldu_matrix* m, int level, int32_t* restrictAddr, double* diag, double* upper,
                    double* lower);

// matrix helpers (ldu_capi.cpp)
int matrix_alloc(ldu_addr* a, ldu_matrix** out);
int matrix_refresh_layout(ldu_matrix* m, hipStream_t onStream = nullptr);   // LDU-space device coefficients -> compute layout
void matrix_free(ldu_matrix* m);
int addr_create_internal(ldu_ctx* ctx, ldu_addr** out, int nCells, int nFaces, const int* l, const int* u);

#if defined(__HIPCC__)
// ---- IEEE double division with the denominator's half done ahead of time.
// `acc / d` is the last operation of every GaussSeidel row (GaussSeidelSmoother.C:154: `curPsi /= diagPtr[cellI]`) and
// sits on the dependency chain of the sweep.  The compiler's f64 division (gfx950 ISA of `a / b`) is
//     s = div_scale(d, d, t); r = rcp(s); e = fma(-s, r, 1); r = fma(r, e, r); e = fma(-s, r, 1); r = fma(r, e, r);
//     n = div_scale(t, d, t); q = n * r; f = fma(-s, q, n); q = div_fmas(f, r, q); div_fixup(q, d, t)
// eleven dependent instructions after t arrives (v_div_scale makes even the reciprocal depend on t).  div_scale only
// rescales - and div_fmas / div_fixup only act - when an exponent is extreme: with both biased exponents in
// [700, 1300] (2^-323 .. 2^277: no zero, denormal, infinity, NaN; |exp(t) - exp(d)| <= 600 < 768) they are the
// identity, and the quotient is `q = t * r; f = fma(-d, q, t); fma(f, r, q)` with r from the SAME five instructions -
// which depend on d alone and are issued before the wait.  Three dependent instructions after t; the same bits
// (tests: every bit-exact sweep test and tools/fuzz_gpu.py run through it; ldu_debug_div_check compares it with
// the compiler's division on random and boundary operands).  Anything outside the range takes the compiler's division.
__device__ __forceinline__ bool ldu_div_exp_safe(double x)
{
    const unsigned e = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;
    return e - 700u <= 600u;
}
// refined reciprocal of d, or 0 when d is outside the safe range (marks "use the plain division")
__device__ __forceinline__ double ldu_div_prepare(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return ldu_div_exp_safe(d) ? r : 0.0;
}
// t / d for the lanes that are active here (the choice is made per wave: no divergence)
__device__ __forceinline__ double ldu_div(double t, double d, double r)
{
    // the quotient is computed unconditionally (three dependent instructions); the range test runs beside it and
    // only a wave that holds an operand outside the range goes through the compiler's division as well
    const double q = t * r;
    const double f = __builtin_fma(-d, q, t);
    double out = __builtin_fma(f, r, q);
    asm volatile("" : "+v"(out));   // ... before the branch below, not inside it
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!((r != 0.0) & ldu_div_exp_safe(t))) != 0ull, 0))
    {
        // (the volatile statement keeps the compiler from turning the branch into compute-both-and-select)
        asm volatile("; operand outside the range: the compiler's division");
        out = t / d;
    }
    return out;
}

// The abort flag is ONE address.  Read by thread 0 of every workgroup before every task (as in round 1) that is
// ~80 M system-scope loads per second on one memory channel at 216^3 - the same order as the ~70 M/s one address
// takes for atomics (why the ticket counter was split in eight) - and it throttled every ticketed kernel: the finest
// GaussSeidel launch ran at 9.9 us per task and wave, 4 us of it outside the task (tools/cluster_trace.py), whatever was
// done to the ticket path.  Now thread 0 looks every 32nd task, a waiting wave at its 8th poll and then every 256th:
// a sweep that aborted still drains (every wait is bounded and looks at the flag), just not within one task.
#define LDU_ABORT_POLL 8u
// Time bound of a dependency wait (the poll bound above counts polls, and a launch that merely CRAWLS - one gpurun box in
// ten ran the 4-sweep launches of an irregular graph 20-1000x slower, DESIGN.md - never reaches it): a wave that has been
// waiting longer than the budget (wall clock, 100 MHz s_memrealtime; default 200 ms, no healthy launch lasts that long)
// gives up like a wave whose polls ran out - abort flag, the grid drains, run_with_fallback re-runs the operation on
// the level kernels.  The clock is read at the 8th poll of a wait and then every 256th: nothing on the fast path.
// One copy per translation unit (set together by ldu_ctx_set_watchdog): [0] budget in ticks (0 = off), [1] debug stall
// in ticks injected into the first task of every sweep launch (tests).
static __device__ unsigned long long g_wait_budget[2] = {20000000ull, 0ull};
__device__ __forceinline__ bool ldu_wait_expired(unsigned& spins, unsigned spinLimit, volatile int* abortFlag,
                                                 unsigned long long& tw0)
{
    if (++spins > spinLimit) return true;
    if ((spins & 255u) != LDU_ABORT_POLL) return false;
    if (*abortFlag) return true;
    const unsigned long long budget = g_wait_budget[0];
    if (!budget) return false;
    const unsigned long long now = wall_clock64();
    if (!tw0) { tw0 = now; return false; }
    return now - tw0 > budget;
}
// tests: the wave that runs the first task of a launch sits still for g_wait_budget[1] ticks
__device__ __forceinline__ void ldu_debug_stall(bool first)
{
    const unsigned long long st = g_wait_budget[1];
    if (first && st)
    {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < st) __builtin_amdgcn_s_sleep(64);
    }
}
__device__ __forceinline__ bool ldu_abort_seen(volatile int* abortFlag, int it)
{
    return (it & 31) == 31 && *abortFlag != 0;
}

// Between two steps of a recurrence that a wavefront runs through LDS (write this step's values, read them in the
// next): LDS instructions of one wave execute in issue order, so the next step's ds_read sees this step's ds_write
// without waiting for the write to retire - only the compiler must not reorder them.  -DLDU_LDS_WAIT=1 restores the
// explicit `s_waitcnt lgkmcnt(0)` (costs a write round trip per step).
#if defined(LDU_LDS_WAIT) && LDU_LDS_WAIT
#define LDU_STEP_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define LDU_STEP_FENCE() asm volatile("" ::: "memory")
#endif
// Release / acquire fences between the wavefronts of ONE workgroup, for LDS only: the fence's address-space operand keeps
// the global loads a wave has in flight (its prefetches) out of the ordering - a workgroup-scope release fence or atomic
// without it waits for them (s_waitcnt vmcnt(0)) in every step.
#define LDU_LDS_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define LDU_LDS_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
#endif
