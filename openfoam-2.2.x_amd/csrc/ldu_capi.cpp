// extern "C" entry points declared in include/ldugpu.h.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "ldu_internal.hpp"

static thread_local std::string g_err;
void ldu_set_error(const std::string& msg) { g_err = msg; }
std::string ldu_last_error_string() { return g_err; }

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool is_device_ptr(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess)
    {
        (void)hipGetLastError();   // unregistered host memory
        return false;
    }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

void ldu_ctx::profStart(const ldu_addr* a, int cat)
{
    if (!profOn || a != profAddr) return;
    ProfCat& P = prof[cat];
    if (P.used + 2 > 16384) return;
    while (P.ev.size() < P.used + 2)
    {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        P.ev.push_back(e);
    }
    (void)hipEventRecord(P.ev[P.used], stream);
}

void ldu_ctx::profStop(const ldu_addr* a, int cat)
{
    if (!profOn || a != profAddr) return;
    ProfCat& P = prof[cat];
    if (P.used + 2 > 16384 || P.ev.size() < P.used + 2) return;
    (void)hipEventRecord(P.ev[P.used + 1], stream);
    P.used += 2;
}

extern "C" {

const char* ldu_last_error(void) { return g_err.c_str(); }

void ldu_default_controls(ldu_controls* c)
{
    memset(c, 0, sizeof(*c));
    c->solver = LDU_SOLVER_PCG;
    c->preconditioner = LDU_PRE_DIC;
    c->smoother = LDU_SM_GAUSSSEIDEL;
    c->tolerance = 1e-6;          // lduMatrixSolver.C:164-169
    c->relTol = 0;
    c->maxIter = 1000;
    c->nSweeps = 1;               // smoothSolver.C:73
    c->cacheAgglomeration = 0;    // GAMGSolver.C:65-76
    c->nPreSweeps = 0;
    c->preSweepsLevelMultiplier = 1;
    c->maxPreSweeps = 4;
    c->nPostSweeps = 2;
    c->postSweepsLevelMultiplier = 1;
    c->maxPostSweeps = 4;
    c->nFinestSweeps = 2;
    c->interpolateCorrection = 0;
    c->scaleCorrection = -1;
    c->directSolveCoarsest = 0;
    c->nCellsInCoarsestLevel = 10;
    c->mergeLevels = 1;
    c->agglomerator = LDU_AGG_FACEAREAPAIR;
    c->nVcycles = 2;              // GAMGPreconditioner.C:60
    c->historyCapacity = 0;
}

// ---------------------------------------------------------------- context

static int ctx_init(ldu_ctx* c, int device);

int ldu_ctx_create(ldu_ctx** out, int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    {
        ldu_set_error("no HIP device available: the lduMatrix GPU path cannot run (no CPU fallback)");
        return -10;
    }
    LDU_CHECK_HIP(hipSetDevice(device));
    ldu_ctx* c = new ldu_ctx();
    c->device = device;
    const int rc = ctx_init(c, device);
    if (rc)
    {
        const std::string why = ldu_last_error_string();
        ldu_ctx_destroy(c);          // (every member is null-safe there: nothing of a half-built context leaks)
        ldu_set_error(why);
        return rc;
    }
    *out = c;
    return 0;
}

static int ctx_init(ldu_ctx* c, int device)
{
    LDU_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    LDU_CHECK_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    LDU_CHECK_HIP(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
    LDU_CHECK_HIP(hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming));
    LDU_CHECK_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    LDU_CHECK_HIP(hipEventCreateWithFlags(&c->evAggFork, hipEventDisableTiming));
    LDU_CHECK_HIP(hipEventCreateWithFlags(&c->evAggJoin, hipEventDisableTiming));
    LDU_CHECK_HIP(hipStreamCreateWithFlags(&c->streamComm, hipStreamNonBlocking));
    LDU_CHECK_HIP(hipEventCreateWithFlags(&c->evPacked, hipEventDisableTiming));
    LDU_CHECK_HIP(hipEventCreateWithFlags(&c->evHalo, hipEventDisableTiming));
    LDU_CHECK_HIP(hipMalloc((void**)&c->d_partials, sizeof(double) * 2 * (size_t)c->maxRedBlocks));
    // the abort flag lives behind the scalar slots so that ONE device-to-host copy brings both
    LDU_CHECK_HIP(hipMalloc((void**)&c->d_scalars, sizeof(double) * (S_NSLOTS + 2)));
    LDU_CHECK_HIP(ldu_memset_sync(c->d_scalars, 0, sizeof(double) * (S_NSLOTS + 2)));
    LDU_CHECK_HIP(hipHostMalloc((void**)&c->h_scalars, sizeof(double) * (S_NSLOTS + 2), hipHostMallocDefault));
    for (int i = 0; i < 2; i++)
    {
        LDU_CHECK_HIP(hipHostMalloc((void**)&c->h_ring[i], sizeof(double) * (S_NSLOTS + 2), hipHostMallocDefault));
        LDU_CHECK_HIP(hipEventCreateWithFlags(&c->evRing[i], hipEventDisableTiming));
    }
    c->d_abort = (int*)(c->d_scalars + S_NSLOTS);
    c->h_abort = (int*)(c->h_scalars + S_NSLOTS);
    *c->h_abort = 0;
    LDU_CHECK_HIP(hipDeviceSynchronize());
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            c->numCUs = prop.multiProcessorCount;
    }
    const char* e = getenv("LDU_SWEEP");
    if (e && !strcmp(e, "levels")) c->sweepP2P = 0;
    e = getenv("LDU_P2P_MAXBPC");
    if (e && atoi(e) > 0) c->p2pMaxBlocksPerCU = atoi(e);
    e = getenv("LDU_GSM_WIDE");
    if (e && atof(e) > 0) c->gsmWideSlices = atof(e);
    e = getenv("LDU_P2P_GATE");
    if (e) c->p2pGate = atoi(e);
    e = getenv("LDU_DUAL_STREAM");
    if (e) c->dualStream = atoi(e);
    e = getenv("LDU_GS_MAXSKEW");
    if (e) c->gsPipelineMaxSkew = atoi(e);
    e = getenv("LDU_GS_PIPELINE");
    if (e) c->gsPipeline = atoi(e);
    e = getenv("LDU_P2P_BPC");
    if (e && atoi(e) > 0) { c->p2pBlocksPerCU = atoi(e); c->p2pBpcForced = 1; }
    e = getenv("LDU_P2P_BACKOFF");
    if (e) k_set_p2p_backoff((unsigned)atoi(e));
    e = getenv("LDU_P2P_BACKOFF_CAP");
    if (e) k_set_p2p_backoff_cap((unsigned)atoi(e));
    e = getenv("LDU_P2P_SLEEP");
    if (e) k_set_p2p_sleep(atoi(e));
    e = getenv("LDU_NO_GRAPH");
    if (e && atoi(e)) c->useGraphs = false;
    e = getenv("LDU_FUSE_ROWS");
    if (e) c->fuseRows = atoi(e);
    e = getenv("LDU_CLUSTER");
    if (e) c->clusterEngine = atoi(e);
    e = getenv("LDU_CLUSTER_BPC_MULTI");
    if (e && atoi(e) > 0) c->clusterBlocksPerCUMulti = atoi(e);
    e = getenv("LDU_CLUSTER_MULTI");
    if (e) c->clusterMulti = atoi(e);
    e = getenv("LDU_CLUSTER_PAYS");
    if (e) c->clusterPaysFactor = atof(e);
    e = getenv("LDU_CLUSTER_MIN");
    if (e) c->clusterMinCells = atoi(e);
    e = getenv("LDU_CLUSTER_BPC");
    if (e && atoi(e) > 0) { c->clusterBlocksPerCU = atoi(e); c->clusterBpcForced = 1; }
    e = getenv("LDU_SMALL");
    if (e) c->smallKernels = atoi(e);
    e = getenv("LDU_STAGE_OVERLAP");
    if (e) c->stageOverlap = atoi(e);
    e = getenv("LDU_CLUSTER_DIRECT");
    if (e) c->clusterDirectFill = atoi(e);
    e = getenv("LDU_WG");
    if (e) c->wgEngine = atoi(e);
    e = getenv("LDU_WG_MAX");
    if (e) c->wgMaxCells = atoi(e);
    e = getenv("LDU_WG_MIN");
    if (e) c->wgMinCells = atoi(e);
    e = getenv("LDU_WG_WIDE");
    if (e) c->wgWide = atoi(e);
    e = getenv("LDU_WG_WAVES");
    if (e) c->wgWaves = atoi(e);
    e = getenv("LDU_BLK");
    if (e) c->blkEngine = atoi(e);
    e = getenv("LDU_BLK_MIN");
    if (e) c->blkMinCells = atoi(e);
    e = getenv("LDU_BLK_MAX");
    if (e) c->blkMaxCells = atoi(e);
    e = getenv("LDU_BLK_EQUAL_MAX");
    if (e) c->blkEqualMax = atoi(e);
    e = getenv("LDU_BLK_BY_SLOTS");
    if (e) c->blkBySlots = atoi(e);
    e = getenv("LDU_BLK_CELLS");
    if (e) c->blkCells = atoi(e);
    e = getenv("LDU_BLK_CELLS_MIN");
    if (e && atoi(e) > 0) c->blkCellsMin = atoi(e);
    e = getenv("LDU_BLK_CELLS_MAX");
    if (e && atoi(e) > 0) c->blkCellsMax = atoi(e);
    e = getenv("LDU_BLK_WAVES");
    if (e) c->blkWaves = atoi(e);
    e = getenv("LDU_KRYLOV_SPECULATE");
    if (e) c->krylovSpeculate = atoi(e);
    e = getenv("LDU_BLK_WIDE_FROM");
    if (e) c->blkWideFrom = atoi(e);
    e = getenv("LDU_BLK_WPS");
    if (e) c->blkWavesPerSweep = atoi(e);
    e = getenv("LDU_BLK_XCD");
    if (e) c->blkXcdMap = atoi(e);
    e = getenv("LDU_GS_LAYOUTS");
    if (e) c->gsLayouts = atoi(e) != 0;
    e = getenv("LDU_GS_LAYOUTS_MIN");
    if (e) c->gsLayoutsMinCells = atoi(e);
    e = getenv("LDU_BLK_LAYOUTS");
    if (e && atoi(e) > 0) c->blkLayouts = atoi(e);
    e = getenv("LDU_BLK_PER_CU");
    if (e && atoi(e) > 0) c->blkMaxPerCU = atoi(e);
    e = getenv("LDU_DEVICE_SHARERS");
    if (e && atoi(e) > 0) c->deviceSharers = atoi(e);
    e = getenv("LDU_SMALL_MAX");
    if (e) c->smallMaxCells = std::min(atoi(e), 16384);
    e = getenv("LDU_P2P_WIDE");
    if (e) k_set_p2p_wide(atoi(e));
    e = getenv("LDU_GS_WIDE_UPPER");
    if (e) c->gsWideUpper = atoi(e);
    e = getenv("LDU_SMALL_PIPE");
    if (e) c->smallPipe = atoi(e);
    e = getenv("LDU_P2P_SLABS");
    if (e) c->p2pSlabs = atoi(e);
    e = getenv("LDU_P2P_WINDOW");
    if (e) c->p2pWindowLevels = atof(e);
    e = getenv("LDU_AGG_OVERLAP");
    if (e) c->aggOverlap = atoi(e);
    e = getenv("LDU_AGG_PREFILL");
    if (e) c->aggPrefill = atoi(e);
    e = getenv("LDU_HALO_OVERLAP");
    if (e) c->haloOverlap = atoi(e);
    e = getenv("LDU_WATCHDOG_MS");
    if (e && (k_set_watchdog((unsigned long long)(atof(e) * 1e5), 0) || k_cluster_set_watchdog((unsigned long long)(atof(e) * 1e5), 0) ||
              k_blocks_set_watchdog((unsigned long long)(atof(e) * 1e5), 0)))
        return -1;
    e = getenv("LDU_LAG_BUCKETS");
    if (e) c->lagBucketWidth = atoi(e);
    e = getenv("LDU_COOP_ROWS");
    if (e) c->coopRows = atoi(e);
    e = getenv("LDU_SORT_ROWS");
    if (e) c->sortRowsByWidth = atoi(e);
    e = getenv("LDU_SPIN_LIMIT");
    if (e && (k_set_spin_limit((unsigned)strtoul(e, nullptr, 10)) || k_cluster_set_spin_limit((unsigned)strtoul(e, nullptr, 10))))
        return -1;
    if (c->sweepP2P && c->p2pSlabs != 0 && k_xcd_census(c)) return -1;
    return 0;
}

int ldu_ctx_destroy(ldu_ctx* c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_destroy(c);
    (void)hipFree(c->d_partials);
    (void)hipFree(c->d_scalars);
    (void)hipHostFree(c->h_scalars);
    for (int i = 0; i < 2; i++)
    {
        if (c->h_ring[i]) (void)hipHostFree(c->h_ring[i]);
        if (c->evRing[i]) (void)hipEventDestroy(c->evRing[i]);
    }
    if (c->evFork) (void)hipEventDestroy(c->evFork);
    if (c->evJoin) (void)hipEventDestroy(c->evJoin);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    if (c->evAggFork) (void)hipEventDestroy(c->evAggFork);
    if (c->evAggJoin) (void)hipEventDestroy(c->evAggJoin);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->streamComm) (void)hipStreamSynchronize(c->streamComm);
    if (c->evPacked) (void)hipEventDestroy(c->evPacked);
    if (c->evHalo) (void)hipEventDestroy(c->evHalo);
    if (c->streamComm) (void)hipStreamDestroy(c->streamComm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    delete c;
    return 0;
}

int ldu_ctx_set_spin_limit(ldu_ctx* c, uint32_t polls)
{
    LDU_CHECK_HIP(hipSetDevice(c->device));
    LDU_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (k_set_spin_limit(polls) || k_cluster_set_spin_limit(polls)) return -1;
    return 0;
}

int ldu_ctx_set_watchdog(ldu_ctx* c, double budgetMs, double debugStallMs)
{
    if (!c || budgetMs < 0 || debugStallMs < 0) { ldu_set_error("ldu_ctx_set_watchdog: bad argument"); return -1; }
    LDU_CHECK_HIP(hipSetDevice(c->device));
    LDU_CHECK_HIP(hipStreamSynchronize(c->stream));
    LDU_CHECK_HIP(hipStreamSynchronize(c->stream2));
    // wall_clock64() ticks at 100 MHz on gfx950 (s_memrealtime)
    const unsigned long long b = (unsigned long long)(budgetMs * 1e5), st = (unsigned long long)(debugStallMs * 1e5);
    if (k_set_watchdog(b, st) || k_cluster_set_watchdog(b, st) || k_blocks_set_watchdog(b, st)) return -1;
    return 0;
}

int ldu_ctx_comm_counters(const ldu_ctx* c, int64_t out[4])
{
    if (!c || !out) { ldu_set_error("ldu_ctx_comm_counters: null argument"); return -1; }
    out[0] = c->nHaloExchanges; out[1] = c->nAllReduces; out[2] = c->nScalarReadbacks; out[3] = c->nHaloOverlapped;
    return 0;
}

int64_t ldu_ctx_fallback_count(const ldu_ctx* c) { return c ? (int64_t)c->nFallbacks : 0; }
int64_t ldu_ctx_overlapped_halo_count(const ldu_ctx* c) { return c ? (int64_t)c->nHaloOverlapped : 0; }

int ldu_ctx_sync(ldu_ctx* c)
{
    LDU_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (c->stream3) LDU_CHECK_HIP(hipStreamSynchronize(c->stream3));
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- addressing

int addr_create_internal(ldu_ctx* ctx, ldu_addr** out, int nCells, int nFaces, const int* l, const int* u)
{
    ldu_addr* a = new ldu_addr();
    a->ctx = ctx;
    a->nCells = nCells;
    a->nFaces = nFaces;
    a->l.assign(l, l + nFaces);
    a->u.assign(u, u + nFaces);
    int rc = plan_build(a);
    if (rc) { plan_free(a); delete a; return rc; }
    a->finalized = true;   // no patches unless added
    *out = a;
    return 0;
}

extern "C" {

int ldu_addr_create(ldu_ctx* ctx, ldu_addr** a, int32_t nCells, int32_t nFaces, const int32_t* lowerAddr,
                    const int32_t* upperAddr)
{
    if (!ctx) { ldu_set_error("null context"); return -11; }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    std::vector<int> l(nFaces), u(nFaces);
    if (nFaces)
    {
        LDU_CHECK_HIP(hipMemcpy(l.data(), lowerAddr, sizeof(int) * nFaces, hipMemcpyDefault));
        LDU_CHECK_HIP(hipMemcpy(u.data(), upperAddr, sizeof(int) * nFaces, hipMemcpyDefault));
    }
    return addr_create_internal(ctx, a, nCells, nFaces, l.data(), u.data());
}

int ldu_addr_add_patch(ldu_addr* a, int32_t n, const int32_t* faceCells, int32_t nbrRank)
{
    Patch p;
    p.n = n;
    p.nbrRank = nbrRank;
    p.faceCells.assign(faceCells, faceCells + n);
    for (int i = 0; i < n; i++)
        if (faceCells[i] < 0 || faceCells[i] >= a->nCells) { ldu_set_error("patch faceCells out of range"); return -12; }
    a->patches.push_back(p);
    a->finalized = false;
    return 0;
}

int ldu_addr_sweep_engine(ldu_addr* a, int32_t kind)
{
    if (kind < 0 || kind > 2) { ldu_set_error("ldu_addr_sweep_engine: kind must be 0..2"); return -2; }
    addr_bg_wait(a);      // (the engine the addressing ends up on, not the stand-in while its plans are being built)
    return k_engine_of(a, kind);
}

int ldu_addr_add_cyclic_patch(ldu_addr* a, int32_t n, const int32_t* faceCells, int32_t nbrPatch)
{
    if (nbrPatch < 0) { ldu_set_error("ldu_addr_add_cyclic_patch: nbrPatch must be >= 0"); return -12; }
    const int rc = ldu_addr_add_patch(a, n, faceCells, -1);
    if (rc) return rc;
    a->patches.back().nbrPatch = nbrPatch;
    return 0;
}

int ldu_addr_finalize(ldu_addr* a)
{
    for (size_t p = 0; p < a->patches.size(); p++)
    {
        const int q = a->patches[p].nbrPatch;
        if (q < 0) continue;
        if (q >= (int)a->patches.size() || q == (int)p || a->patches[q].nbrPatch != (int)p
            || a->patches[q].n != a->patches[p].n)
        {
            ldu_set_error("ldu_addr_finalize: cyclic patch " + std::to_string(p) + " has no matching neighbour patch");
            return -12;
        }
    }
    return plan_finalize_patches(a);
}

int ldu_addr_destroy(ldu_addr* a)
{
    if (!a) return 0;
    (void)hipStreamSynchronize(a->ctx->stream);
    plan_free(a);
    delete a;
    return 0;
}

int ldu_addr_info(const ldu_addr* a, int32_t* nLevels, int32_t* nSlices, int64_t* nEntriesPadded)
{
    if (nLevels) *nLevels = a->nLevels;
    if (getenv("LDU_VERBOSE"))
        fprintf(stderr, "[ldugpu] addressing: %d cells, %d levels, %d slices, %d XCD slabs\n", a->nCells,
                a->nLevels, a->nSlices, a->nSlabs);
    if (nSlices) *nSlices = a->nSlices;
    if (nEntriesPadded) *nEntriesPadded = a->nEntries;
    return 0;
}

int ldu_addr_set_face_weights(ldu_addr* a, const double* w)
{
    a->faceWeights.resize(a->nFaces);
    if (a->nFaces) LDU_CHECK_HIP(hipMemcpy(a->faceWeights.data(), w, sizeof(double) * a->nFaces, hipMemcpyDefault));
    return 0;
}

int ldu_addr_set_subdomains(ldu_addr* a, int32_t nSub, const int32_t* cellSub)
{
    if (nSub <= 0) { a->subOf.clear(); a->nSub = 0; return 0; }
    a->subOf.assign(cellSub, cellSub + a->nCells);
    for (int c = 0; c < a->nCells; c++)
        if (a->subOf[c] < 0 || a->subOf[c] >= nSub) { a->subOf.clear(); a->nSub = 0; ldu_set_error("ldu_addr_set_subdomains: label out of range"); return -2; }
    // a sub-domain is a rank: internal faces never cross (what couples two of them is an interface)
    for (int f = 0; f < a->nFaces; f++)
        if (a->subOf[a->l[f]] != a->subOf[a->u[f]])
        {
            a->subOf.clear(); a->nSub = 0;
            ldu_set_error("ldu_addr_set_subdomains: an internal face joins two sub-domains (couple them through cyclic patches)");
            return -2;
        }
    a->nSub = nSub;
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- matrix

int matrix_alloc(ldu_addr* a, ldu_matrix** out)
{
    ldu_matrix* m = new ldu_matrix();
    m->a = a;
    const size_t nC = (size_t)a->nCells + 1, nF = (size_t)a->nFaces + 1;
    const size_t nE = (size_t)(a->nEntries > 0 ? a->nEntries : 1);
    LDU_CHECK_HIP(hipMalloc((void**)&m->d_diagO, sizeof(double) * nC));
    LDU_CHECK_HIP(hipMalloc((void**)&m->d_upperO, sizeof(double) * nF));
    m->d_lowerO = m->d_upperO;
    LDU_CHECK_HIP(hipMalloc((void**)&m->d_diag, sizeof(double) * nC));
    LDU_CHECK_HIP(hipMalloc((void**)&m->d_valA, sizeof(double) * nE));
    m->d_valT = m->d_valA;
    if (a->nPatchFaces)
    {
        LDU_CHECK_HIP(hipMalloc((void**)&m->d_bou, sizeof(double) * (size_t)a->nPatchFaces));
        LDU_CHECK_HIP(hipMalloc((void**)&m->d_int, sizeof(double) * (size_t)a->nPatchFaces));
        LDU_CHECK_HIP(ldu_memset_sync(m->d_bou, 0, sizeof(double) * (size_t)a->nPatchFaces));
        LDU_CHECK_HIP(ldu_memset_sync(m->d_int, 0, sizeof(double) * (size_t)a->nPatchFaces));
    }
    *out = m;
    return 0;
}

void matrix_free(ldu_matrix* m)
{
    if (!m) return;
    if (m->gamg) gamg_free(m->gamg);
    coupled_free(m);
    coarsest_lu_free(m);
    // the cluster engine keeps converted copies of this matrix's value arrays, keyed by their addresses
    for (const double* v : {(const double*)m->d_valA, (const double*)m->d_valT, (const double*)m->d_valP,
                            (const double*)m->d_valPT})
        if (v) { cluster_forget(m->a, v); blocks_forget(m->a, v); gs_layouts_forget(m->a, v); m->a->valOrigin.erase(v); m->a->arrStamp.erase(v); }
    if (m->d_lowerO && m->d_lowerO != m->d_upperO) (void)hipFree(m->d_lowerO);
    if (m->d_valT && m->d_valT != m->d_valA) (void)hipFree(m->d_valT);
    void* ptrs[] = {m->d_diagO, m->d_upperO, m->d_diag, m->d_valA, m->d_bou, m->d_int, m->d_rD,
                    m->d_valP, m->d_valPT, m->d_rDiag};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (double* p : m->work) if (p) (void)hipFree(p);
    delete m;
}

// LDU-space device coefficients (original order) -> level-ordered diag + sliced-ELL values.
int matrix_refresh_layout(ldu_matrix* m, hipStream_t onStream)
{
    ldu_addr* a = m->a;
    hipStream_t s = onStream ? onStream : a->ctx->stream;
    const size_t nE = (size_t)(a->nEntries > 0 ? a->nEntries : 1);
    if (k_permute_in(a, m->d_diag, m->d_diagO, s)) return -1;
    if (k_fill_sell(a, m->d_lowerO, m->d_upperO, m->d_valA, s)) return -1;
    a->valOrigin[m->d_valA] = {m->d_lowerO, m->d_upperO};
    if (!m->sym)
    {
        if (m->d_valT == m->d_valA) LDU_CHECK_HIP(hipMalloc((void**)&m->d_valT, sizeof(double) * nE));
        if (k_fill_sell(a, m->d_upperO, m->d_lowerO, m->d_valT, s)) return -1;
        a->valOrigin[m->d_valT] = {m->d_upperO, m->d_lowerO};
    }
    else if (m->d_valT != m->d_valA)
    {
        cluster_forget(a, m->d_valT);
        blocks_forget(a, m->d_valT);
        gs_layouts_forget(a, m->d_valT);
        a->valOrigin.erase(m->d_valT);
        a->arrStamp.erase(m->d_valT);
        (void)hipFree(m->d_valT);
        m->d_valT = m->d_valA;
    }
    m->rDKind = -1;
    m->rDiagValid = false;
    m->coeffEpoch++;
    m->haveCoeffs = true;
    return 0;
}

extern "C" {

int ldu_matrix_create(ldu_addr* a, ldu_matrix** m)
{
    if (!a->finalized) { ldu_set_error("ldu_addr_finalize() must follow ldu_addr_add_patch()"); return -13; }
    LDU_CHECK_HIP(hipSetDevice(a->ctx->device));
    return matrix_alloc(a, m);
}

int ldu_matrix_destroy(ldu_matrix* m)
{
    if (!m) return 0;
    (void)hipStreamSynchronize(m->a->ctx->stream);
    matrix_free(m);
    return 0;
}

int ldu_matrix_set_coeffs(ldu_matrix* m, const double* diag, const double* upper, const double* lower)
{
    ldu_addr* a = m->a;
    hipStream_t s = a->ctx->stream;
    LDU_CHECK_HIP(hipSetDevice(a->ctx->device));
    const bool sym = (lower == nullptr) || (lower == upper);
    // (a coefficient agglomeration of the previous coefficients that nobody joined yet - a hierarchy built for a query, a
    //  solve that failed - still reads the LDU arrays on ctx->stream3: the copies below wait for it; no-op otherwise)
    if (a->ctx->evAggJoin) LDU_CHECK_HIP(hipStreamWaitEvent(s, a->ctx->evAggJoin, 0));
    if (!sym && m->d_lowerO == m->d_upperO)
        LDU_CHECK_HIP(hipMalloc((void**)&m->d_lowerO, sizeof(double) * ((size_t)a->nFaces + 1)));
    if (sym && m->d_lowerO != m->d_upperO)
    {
        (void)hipFree(m->d_lowerO);
        m->d_lowerO = m->d_upperO;
    }
    m->sym = sym;
    LDU_CHECK_HIP(hipMemcpyAsync(m->d_diagO, diag, sizeof(double) * a->nCells, hipMemcpyDefault, s));
    if (a->nFaces)
    {
        LDU_CHECK_HIP(hipMemcpyAsync(m->d_upperO, upper, sizeof(double) * a->nFaces, hipMemcpyDefault, s));
        if (!sym) LDU_CHECK_HIP(hipMemcpyAsync(m->d_lowerO, lower, sizeof(double) * a->nFaces, hipMemcpyDefault, s));
    }
    // (the LDU arrays are in place: what the coefficient agglomeration of a GAMG hierarchy waits for, gamg.cpp)
    const bool forked = a->ctx->evAggFork && hipEventRecord(a->ctx->evAggFork, s) == hipSuccess;
    const int rc = matrix_refresh_layout(m);
    a->ctx->aggForkOf = forked ? m : nullptr;
    a->ctx->aggForkEpoch = m->coeffEpoch;
    // host arrays belong to the caller again when this returns (an asynchronous copy from pageable memory may still be
    // reading them: the runtime pins the pages and returns); device arrays are only read by work already ordered on the stream
    if (!is_device_ptr(diag) || (a->nFaces && (!is_device_ptr(upper) || (!sym && !is_device_ptr(lower)))))
        LDU_CHECK_HIP(hipStreamSynchronize(s));
    return rc;
}

int ldu_matrix_set_patch_coeffs(ldu_matrix* m, int32_t patchI, const double* bou, const double* intc)
{
    ldu_addr* a = m->a;
    if (patchI < 0 || patchI >= (int)a->patches.size()) { ldu_set_error("bad patch index"); return -14; }
    const Patch& p = a->patches[patchI];
    hipStream_t s = a->ctx->stream;
    if (p.n)
    {
        if (a->ctx->evAggJoin) LDU_CHECK_HIP(hipStreamWaitEvent(s, a->ctx->evAggJoin, 0));   // (as in set_coeffs)
        LDU_CHECK_HIP(hipMemcpyAsync(m->d_bou + p.offset, bou, sizeof(double) * p.n, hipMemcpyDefault, s));
        LDU_CHECK_HIP(hipMemcpyAsync(m->d_int + p.offset, intc, sizeof(double) * p.n, hipMemcpyDefault, s));
        if (!is_device_ptr(bou) || !is_device_ptr(intc)) LDU_CHECK_HIP(hipStreamSynchronize(s));   // (as in set_coeffs)
    }
    m->coeffEpoch++;   // coarse-level interface coefficients follow
    // (layouts that hold interface coefficients next to the matrix coefficients: ldu_blocks.hip)
    if (m->d_valA) val_touch(a, m->d_valA);
    if (m->d_valT && m->d_valT != m->d_valA) val_touch(a, m->d_valT);
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- vector staging
// User vectors are in the ORIGINAL cell order, host or device.  in(): -> device, level order.

struct Stager {
    ldu_matrix* m;
    ldu_addr* a;
    hipStream_t s;
    int nextScratch = 0, nextWork = 20;
    explicit Stager(ldu_matrix* mm) : m(mm), a(mm->a), s(mm->a->ctx->stream) {}

    // returns device pointer in level order
    double* in(const double* user)
    {
        const double* devOrig = user;
        if (!is_device_ptr(user))
        {
            double* st = a->scratchVec(nextScratch++);
            if (!st) return nullptr;
            // host vectors travel on the second stream: device work already queued on the main stream (the GAMG coarse
            // matrices of this solve, ldu_solve) runs while they cross PCIe; the main stream waits for the copy only
            ldu_ctx* ctx = a->ctx;
            hipStream_t sc = ctx->stageOverlap ? ctx->stream2 : s;
            if (a->nCells && hipMemcpyAsync(st, user, sizeof(double) * a->nCells, hipMemcpyHostToDevice, sc) != hipSuccess)
                return nullptr;
            if (sc != s)
            {
                if (hipEventRecord(ctx->evFork, sc) != hipSuccess || hipStreamWaitEvent(s, ctx->evFork, 0) != hipSuccess)
                    return nullptr;
            }
            devOrig = st;
        }
        double* w = m->workVec(nextWork++);
        if (!w) return nullptr;
        if (a->nCells && k_permute_in(a, w, devOrig, s)) return nullptr;
        return w;
    }
    double* tmp() { return m->workVec(nextWork++); }
    int out(double* user, const double* devNew)
    {
        if (!a->nCells) return 0;
        if (is_device_ptr(user))
        {
            // blocking like the reference's solve(): the caller may touch `user` from any
            // stream as soon as we return
            if (k_permute_out(a, user, devNew, s)) return -1;
            LDU_CHECK_HIP(hipStreamSynchronize(s));
            return 0;
        }
        double* st = a->scratchVec(nextScratch++);
        if (!st) return -1;
        if (k_permute_out(a, st, devNew, s)) return -1;
        LDU_CHECK_HIP(hipMemcpyAsync(user, st, sizeof(double) * a->nCells, hipMemcpyDeviceToHost, s));
        LDU_CHECK_HIP(hipStreamSynchronize(s));
        return 0;
    }
};

#define NEED_COEFFS(m)                                                                    \
    do {                                                                                  \
        if (!(m) || !(m)->haveCoeffs) { ldu_set_error("ldu_matrix_set_coeffs() not called"); return -15; } \
        LDU_CHECK_HIP(hipSetDevice((m)->a->ctx->device));                                 \
    } while (0)

extern "C" {

int ldu_amul(ldu_matrix* m, double* Apsi, const double* psi)
{
    NEED_COEFFS(m);
    Stager S(m);
    double* x = S.in(psi);
    double* y = S.tmp();
    if (!x || !y) return -1;
    if (dev_amul(m, y, x, false)) return -1;
    return S.out(Apsi, y);
}

int ldu_tmul(ldu_matrix* m, double* Tpsi, const double* psi)
{
    NEED_COEFFS(m);
    Stager S(m);
    double* x = S.in(psi);
    double* y = S.tmp();
    if (!x || !y) return -1;
    if (dev_amul(m, y, x, true)) return -1;
    return S.out(Tpsi, y);
}

int ldu_sumA(ldu_matrix* m, double* sumA)
{
    NEED_COEFFS(m);
    Stager S(m);
    double* y = S.tmp();
    if (!y) return -1;
    if (dev_sumA(m, y)) return -1;
    return S.out(sumA, y);
}

int ldu_residual(ldu_matrix* m, double* rA, const double* psi, const double* source)
{
    NEED_COEFFS(m);
    Stager S(m);
    double* x = S.in(psi);
    double* b = S.in(source);
    double* y = S.tmp();
    if (!x || !b || !y) return -1;
    if (dev_residual(m, y, x, b)) return -1;
    return S.out(rA, y);
}

int ldu_H(ldu_matrix* m, double* H, const double* psi)
{
    NEED_COEFFS(m);
    Stager S(m);
    double* x = S.in(psi);
    double* y = S.tmp();
    if (!x || !y) return -1;
    if (k_offdiag(m, y, x, 0, S.s)) return -1;
    return S.out(H, y);
}

int ldu_H1(ldu_matrix* m, double* H1)
{
    NEED_COEFFS(m);
    Stager S(m);
    double* y = S.tmp();
    if (!y) return -1;
    if (k_offdiag(m, y, nullptr, 1, S.s)) return -1;
    return S.out(H1, y);
}

int ldu_faceH(ldu_matrix* m, double* faceH, const double* psi)
{
    NEED_COEFFS(m);
    ldu_addr* a = m->a;
    hipStream_t s = a->ctx->stream;
    const double* x = psi;
    if (!is_device_ptr(psi))
    {
        double* st = a->scratchVec(0);
        LDU_CHECK_HIP(hipMemcpyAsync(st, psi, sizeof(double) * a->nCells, hipMemcpyHostToDevice, s));
        x = st;
    }
    double* out = is_device_ptr(faceH) ? faceH : a->scratchVec(1);
    if (k_faceH(m, out, x, s)) return -1;
    if (out != faceH)
    {
        LDU_CHECK_HIP(hipMemcpyAsync(faceH, out, sizeof(double) * a->nFaces, hipMemcpyDeviceToHost, s));
    }
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

int ldu_gSumProd(ldu_matrix* m, const double* a_, const double* b_, double* result)
{
    NEED_COEFFS(m);
    Stager S(m);
    ldu_ctx* ctx = m->a->ctx;
    double* x = S.in(a_);
    double* y = S.in(b_);
    if (!x || !y) return -1;
    if (k_reduce(ctx, m->a->nCells, RED_DOT, x, y, nullptr, nullptr, S_TMP0, S.s)) return -1;
    if (comm_allreduce_scalars(ctx, S_TMP0, 1, S.s)) return -1;
    return dev_read_scalars(ctx, S_TMP0, 1, result);
}

int ldu_gSumMag(ldu_matrix* m, const double* a_, double* result)
{
    NEED_COEFFS(m);
    Stager S(m);
    ldu_ctx* ctx = m->a->ctx;
    double* x = S.in(a_);
    if (!x) return -1;
    if (k_reduce(ctx, m->a->nCells, RED_SUMMAG, x, nullptr, nullptr, nullptr, S_TMP0, S.s)) return -1;
    if (comm_allreduce_scalars(ctx, S_TMP0, 1, S.s)) return -1;
    return dev_read_scalars(ctx, S_TMP0, 1, result);
}

int ldu_precondition(ldu_matrix* m, int32_t pre, double* wA, const double* rA, int32_t transpose)
{
    NEED_COEFFS(m);
    if (!m->sym && (pre == LDU_PRE_DIC || pre == LDU_PRE_FDIC))
    {
        ldu_set_error("DIC/FDIC are symmetric-matrix preconditioners (lduMatrixPreconditioner.C:98-124)");
        return -16;
    }
    if (pre == LDU_PRE_GAMG) { ldu_set_error("use ldu_solve with preconditioner GAMG"); return -16; }
    return run_with_fallback(m, [&]() -> int {
        Stager S(m);
        double* r = S.in(rA);
        double* w = S.tmp();
        if (!r || !w) return -1;
        if (dev_precondition(m, pre, w, r, transpose != 0, S.s)) return -1;
        if (int rc = dev_check_abort(m->a->ctx)) return rc;
        return S.out(wA, w);
    });
}

int ldu_smooth(ldu_matrix* m, int32_t smoother, double* psi, const double* source, int32_t nSweeps)
{
    NEED_COEFFS(m);
    return run_with_fallback(m, [&]() -> int {
        Stager S(m);
        double* x = S.in(psi);
        double* b = S.in(source);
        if (!x || !b) return -1;
        if (dev_smooth(m, smoother, x, b, nSweeps)) return -1;
        if (int rc = dev_check_abort(m->a->ctx)) return rc;
        return S.out(psi, x);
    });
}

int ldu_solve(ldu_matrix* m, const ldu_controls* c, double* psi, const double* source, ldu_perf* perf,
              double* resHistory)
{
    NEED_COEFFS(m);
    return run_with_fallback(m, [&]() -> int {
        memset(perf, 0, sizeof(*perf));
        // (the clock starts here in both cases, so that solveSeconds is comparable between host- and device-pointer calls:
        //  it contains the level-matrix build either way)
        LDU_CHECK_HIP(hipStreamSynchronize(m->a->ctx->stream));
        const double t0 = now_s();
        // GAMG: the coarse matrices of this solve first (device work only), so that host vectors are uploaded meanwhile
        if (m->a->ctx->stageOverlap && (c->solver == LDU_SOLVER_GAMG || c->preconditioner == LDU_PRE_GAMG)
            && (!is_device_ptr(psi) || !is_device_ptr(source)))
            if (gamg_precondition_setup(m, c)) return -1;
        Stager S(m);
        double* x = S.in(psi);
        double* b = S.in(source);
        if (!x || !b) return -1;
        LDU_CHECK_HIP(hipStreamSynchronize(S.s));
        int rc = dev_solve(m, c, x, b, perf, resHistory);
        LDU_CHECK_HIP(hipStreamSynchronize(S.s));
        perf->solveSeconds = now_s() - t0;
        if (rc) return rc;
        if (int rc2 = dev_check_abort(m->a->ctx)) return rc2;
        return S.out(psi, x);
    });
}

int ldu_debug_granule_tags(ldu_matrix* m, int32_t* tags /* nCells, level order */, int32_t* perm)
{
    ldu_addr* a = m->a;
    LDU_CHECK_HIP(hipStreamSynchronize(a->ctx->stream));
    std::vector<uint4> g(a->nCells);
    LDU_CHECK_HIP(hipMemcpy(g.data(), a->p2p[0].d_granule, sizeof(uint4) * a->nCells, hipMemcpyDeviceToHost));
    for (int i = 0; i < a->nCells; i++) { tags[i] = (int)g[i].y; perm[i] = a->level[a->perm[i]]; }
    return 0;
}

int ldu_debug_p2p_records(ldu_matrix* m, int32_t* out /* 1 + 512 */)
{
    LDU_CHECK_HIP(hipStreamSynchronize(m->a->ctx->stream));
    return k_read_p2p_dbg_records(out);
}

int ldu_debug_cluster_trace(ldu_matrix* m, void* buf)
{
    LDU_CHECK_HIP(hipStreamSynchronize(m->a->ctx->stream));
    return k_cluster_set_trace((unsigned long long*)buf);
}

int ldu_debug_gs_multi_trace(ldu_matrix* m, void* buf)
{
    LDU_CHECK_HIP(hipStreamSynchronize(m->a->ctx->stream));
    return k_set_gs_multi_trace((unsigned long long*)buf, m->a->nSlices);
}

int ldu_debug_blocks_trace(ldu_matrix* m, void* buf)
{
    LDU_CHECK_HIP(hipStreamSynchronize(m->a->ctx->stream));
    return k_blocks_set_trace((unsigned long long*)buf);
}

int ldu_debug_blocks_info(ldu_matrix* m, int32_t k, int64_t* out)
{
    long o[8];
    const int rc = k_blocks_info(m->a, k, o);
    for (int i = 0; i < 8; i++) out[i] = o[i];
    return rc < 0 ? -1 : 0;
}

int ldu_debug_gs_layouts(ldu_matrix* m, int64_t* out)
{
    const ldu_addr* a = m->a;
    for (int i = 0; i < 6; i++) out[i] = 0;
    out[0] = a->gsLayState == 1 ? a->gsLayBuilt : 0;
    for (int j = 1; j < 4; j++) out[1 + j] = a->gsLay[j] ? a->gsLay[j]->nSlices : 0;
    out[5] = a->nSlices;
    return 0;
}

int ldu_debug_slice_levels(ldu_matrix* m, int32_t* out, int32_t cap)
{
    const ldu_addr* a = m->a;
    if (cap < a->nLevels + 2) { ldu_set_error("ldu_debug_slice_levels: buffer too small"); return -1; }
    out[0] = a->nLevels;
    for (int L = 0; L <= a->nLevels; L++) out[1 + L] = a->levelSliceStart[L];
    return 0;
}

// per slice: first row, row count, entries per row (padded width), most lower / upper neighbours of its rows
int ldu_debug_slices(ldu_matrix* m, int32_t* out /* [nSlices][5] */, int32_t cap)
{
    const ldu_addr* a = m->a;
    if (cap < a->nSlices) { ldu_set_error("ldu_debug_slices: buffer too small"); return -1; }
    std::vector<int> row(a->nSlices + 1), cnt(a->nSlices), w(a->nSlices);
    std::vector<unsigned char> nl(a->nCells), nu(a->nCells);
    LDU_CHECK_HIP(hipMemcpy(row.data(), a->d_sliceRow, sizeof(int) * (a->nSlices + 1), hipMemcpyDeviceToHost));
    LDU_CHECK_HIP(hipMemcpy(cnt.data(), a->d_sliceCnt, sizeof(int) * a->nSlices, hipMemcpyDeviceToHost));
    LDU_CHECK_HIP(hipMemcpy(w.data(), a->d_sliceW, sizeof(int) * a->nSlices, hipMemcpyDeviceToHost));
    LDU_CHECK_HIP(hipMemcpy(nl.data(), a->d_nL, a->nCells, hipMemcpyDeviceToHost));
    LDU_CHECK_HIP(hipMemcpy(nu.data(), a->d_nU, a->nCells, hipMemcpyDeviceToHost));
    for (int s = 0; s < a->nSlices; s++)
    {
        int ml = 0, mu = 0;
        for (int i = 0; i < cnt[s]; i++) { ml = std::max(ml, (int)nl[row[s] + i]); mu = std::max(mu, (int)nu[row[s] + i]); }
        out[5 * s] = row[s]; out[5 * s + 1] = cnt[s]; out[5 * s + 2] = w[s]; out[5 * s + 3] = ml; out[5 * s + 4] = mu;
    }
    return 0;
}

int ldu_debug_cluster_levels(ldu_matrix* m, int32_t* out, int32_t cap)
{
    return k_cluster_levels(m->a, out, cap);
}

int ldu_debug_div_check(ldu_ctx* ctx, uint64_t seed, int64_t n, uint64_t* mismatches)
{
    if (!ctx || !mismatches) { ldu_set_error("ldu_debug_div_check: null argument"); return -1; }
    unsigned long long bad = 0;
    const int rc = k_div_check(ctx, (unsigned long long)seed, (long)n, &bad);
    *mismatches = bad;
    return rc;
}

int ldu_debug_stream(ldu_ctx* ctx, int32_t mode, int64_t n, int32_t reps, double* seconds)
{
    if (!ctx || !seconds) { ldu_set_error("ldu_debug_stream: null argument"); return -1; }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    return k_stream(ctx, mode, (long)n, reps, seconds);
}

int ldu_debug_p2p_stuck(ldu_matrix* m, int32_t out[16])
{
    LDU_CHECK_HIP(hipStreamSynchronize(m->a->ctx->stream));
    return k_read_p2p_dbg(out);
}

int ldu_debug_p2p_trace(ldu_matrix* m, void* buf)
{
    LDU_CHECK_HIP(hipStreamSynchronize(m->a->ctx->stream));
    m->a->ctx->p2pTrace = buf != nullptr;
    return k_set_p2p_trace((unsigned long long*)buf);
}

int ldu_profile_begin(ldu_matrix* m)
{
    ldu_ctx* ctx = m->a->ctx;
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& c : ctx->prof) { c.used = 0; c.launches = 0; }
    ctx->profAddr = m->a;
    ctx->profOn = true;
    return 0;
}

int ldu_profile_end(ldu_matrix* m, double ms[8], int64_t counts[8])
{
    ldu_ctx* ctx = m->a->ctx;
    ctx->profOn = false;
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < LDU_PROF_NCAT; c++)
    {
        double tot = 0;
        auto& P = ctx->prof[c];
        for (size_t i = 0; i + 1 < P.used; i += 2)
        {
            float t = 0;
            if (hipEventElapsedTime(&t, P.ev[i], P.ev[i + 1]) == hipSuccess) tot += t;
        }
        ms[c] = tot;
        counts[c] = (int64_t)(P.used / 2);
    }
    return 0;
}

int ldu_gamg_levels(ldu_matrix* m, const ldu_controls* c, int32_t* nLevels, int32_t* nCells, int32_t* nFaces)
{
    NEED_COEFFS(m);
    if (gamg_build_for_query(m, c)) return -1;
    return gamg_query(m, nLevels, nCells, nFaces);
}

int ldu_gamg_level_info(ldu_matrix* m, int32_t level, int32_t info[8])
{
    (void)ldu_matrix_wait_plans(m);
    return gamg_level_info(m, level, info);
}

// sweep plans that are still being built behind the solves (ldu_addr::bgPlan): wait for all of them
int ldu_matrix_wait_plans(ldu_matrix* m)
{
    if (!m) return 0;
    addr_bg_wait(m->a);
    gamg_wait_plans(m->gamg);
    return 0;
}

int ldu_gamg_level_data(ldu_matrix* m, int32_t level, int32_t* restrictAddr, double* diag, double* upper,
                        double* lower)
{
    return gamg_level_data(m, level, restrictAddr, diag, upper, lower);
}

// ---------------------------------------------------------------- fv stencils
// Fields are device or host arrays in the ORIGINAL numbering; results likewise.

struct FvBuf {
    ldu_addr* a;
    hipStream_t s;
    std::vector<void*> owned;
    explicit FvBuf(ldu_addr* aa) : a(aa), s(aa->ctx->stream) {}
    ~FvBuf() { for (void* p : owned) (void)hipFree(p); }
    const double* in(const double* user, size_t n)
    {
        if (!user || is_device_ptr(user)) return user;
        double* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(double) * (n ? n : 1)) != hipSuccess) return nullptr;
        owned.push_back(d);
        (void)hipMemcpyAsync(d, user, sizeof(double) * n, hipMemcpyHostToDevice, s);
        return d;
    }
    double* outBuf(double* user, size_t n)
    {
        if (is_device_ptr(user)) return user;
        double* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(double) * (n ? n : 1)) != hipSuccess) return nullptr;
        owned.push_back(d);
        return d;
    }
    int finish(double* user, double* dev, size_t n)
    {
        if (user != dev) LDU_CHECK_HIP(hipMemcpyAsync(user, dev, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        LDU_CHECK_HIP(hipStreamSynchronize(s));
        return 0;
    }
};

int ldu_fv_interpolate(ldu_addr* a, int32_t nComp, const double* lambdas, const double* vf, double* sf)
{
    FvBuf B(a);
    const double* lam = B.in(lambdas, a->nFaces);
    const double* v = B.in(vf, (size_t)a->nCells * nComp);
    double* o = B.outBuf(sf, (size_t)a->nFaces * nComp);
    if (k_fv_interpolate(a, nComp, lam, v, o, B.s)) return -1;
    return B.finish(sf, o, (size_t)a->nFaces * nComp);
}

int ldu_fvc_surfaceIntegrate(ldu_addr* a, int32_t nComp, const double* ssf, const double* V, double* ivf)
{
    FvBuf B(a);
    const double* f = B.in(ssf, (size_t)a->nFaces * nComp);
    const double* v = B.in(V, a->nCells);
    double* o = B.outBuf(ivf, (size_t)a->nCells * nComp);
    if (k_fv_surfaceIntegrate(a, nComp, f, nullptr, v, o, B.s)) return -1;
    return B.finish(ivf, o, (size_t)a->nCells * nComp);
}

int ldu_fvc_gaussGrad(ldu_addr* a, const double* Sf, const double* ssf, const double* V, double* grad)
{
    FvBuf B(a);
    const double* sfv = B.in(Sf, (size_t)a->nFaces * 3);
    const double* f = B.in(ssf, a->nFaces);
    const double* v = B.in(V, a->nCells);
    double* o = B.outBuf(grad, (size_t)a->nCells * 3);
    if (k_fv_surfaceIntegrate(a, 3, f, sfv, v, o, B.s)) return -1;
    return B.finish(grad, o, (size_t)a->nCells * 3);
}

int ldu_fvc_snGrad(ldu_addr* a, const double* deltaCoeffs, const double* vf, double* ssf)
{
    FvBuf B(a);
    const double* d = B.in(deltaCoeffs, a->nFaces);
    const double* v = B.in(vf, a->nCells);
    double* o = B.outBuf(ssf, a->nFaces);
    if (k_fv_snGrad(a, d, v, o, B.s)) return -1;
    return B.finish(ssf, o, a->nFaces);
}

int ldu_fvm_laplacian(ldu_addr* a, const double* deltaCoeffs, const double* gammaMagSf, double* diag,
                      double* upper)
{
    FvBuf B(a);
    const double* d = B.in(deltaCoeffs, a->nFaces);
    const double* g = B.in(gammaMagSf, a->nFaces);
    double* up = B.outBuf(upper, a->nFaces);
    double* dg = B.outBuf(diag, a->nCells);
    // coefficients and negSumDiag in one pass (in-place arguments: the two-kernel path)
    const int fused = k_fv_coeffs_diag(a, 0, d, g, nullptr, up, dg, B.s);
    if (fused < 0) return -1;
    if (fused > 0)
    {
        if (k_fv_laplacian_coeffs(a->nFaces, d, g, up, B.s)) return -1;
        if (k_fv_negSumDiag(a, up, up, dg, B.s)) return -1;
    }
    if (B.finish(upper, up, a->nFaces)) return -1;
    return B.finish(diag, dg, a->nCells);
}

int ldu_fvm_div(ldu_addr* a, const double* weights, const double* faceFlux, double* diag, double* upper,
                double* lower)
{
    FvBuf B(a);
    const double* w = B.in(weights, a->nFaces);
    const double* phi = B.in(faceFlux, a->nFaces);
    double* up = B.outBuf(upper, a->nFaces);
    double* lo = B.outBuf(lower, a->nFaces);
    double* dg = B.outBuf(diag, a->nCells);
    const int fused = k_fv_coeffs_diag(a, 1, w, phi, lo, up, dg, B.s);
    if (fused < 0) return -1;
    if (fused > 0)
    {
        if (k_fv_div_coeffs(a->nFaces, w, phi, lo, up, B.s)) return -1;
        if (k_fv_negSumDiag(a, lo, up, dg, B.s)) return -1;
    }
    if (B.finish(upper, up, a->nFaces)) return -1;
    if (B.finish(lower, lo, a->nFaces)) return -1;
    return B.finish(diag, dg, a->nCells);
}

}  // extern "C"
