// GAMG: pair agglomeration (host, order-dependent greedy algorithm), level matrices
// (device, rebuilt from the fine coefficients every solve like the reference), V-cycle.
//
// Reference: solvers/GAMG/GAMGSolver.C:44-127, GAMGSolverSolve.C:34-487, GAMGSolverScale.C,
// GAMGSolverAgglomerateMatrix.C, GAMGAgglomerations/pairGAMGAgglomeration/pairGAMGAgglomerate.C,
// GAMGAgglomeration/GAMGAgglomerateLduAddressing.C, GAMGPreconditioner.C.
//
// The agglomeration is sequential and order dependent in the reference (greedy matching in
// cell order), so it stays a host algorithm run once per addressing ("cacheAgglomeration");
// everything executed per solve or per cycle runs on the device.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include "ldu_internal.hpp"

static const int kMaxLevels = 50;   // GAMGAgglomeration.C:75
static const double kSmall = 1e-20;

struct GamgLevel {
    ldu_addr* addr = nullptr;
    ldu_matrix* mat = nullptr;
    int nFineCells = 0, nFineFaces = 0;
    std::vector<int> restrictAddr;       // fine (orig) -> coarse (orig)
    std::vector<int> faceRestrictAddr;   // fine face -> coarse face | -(coarse cell)-1
    // device maps, NEW numbering (V-cycle transfers)
    int* d_childStart = nullptr;
    int* d_child = nullptr;
    int* d_mapNew = nullptr;
    // device maps, ORIGINAL numbering (coefficient agglomeration)
    int* d_cfStart = nullptr;
    int* d_cfFine = nullptr;
    unsigned char* d_cfFlip = nullptr;
    int* d_ccStart = nullptr;
    int* d_ccFine = nullptr;
    int* d_childStartO = nullptr;
    int* d_childO = nullptr;
    double* d_corr = nullptr;
    double* d_src = nullptr;
    int* d_pcStart = nullptr;            // coarse patch face -> fine patch faces (concatenated, asc.)
    int* d_pcFine = nullptr;
    // this level's matrix (coefficients, level layout, engine copies) is being written on ctx->stream3: recorded behind the
    // level's last fill; the main stream waits for it before the level's first use in a V-cycle (gamg_join_level)
    hipEvent_t evReady = nullptr;
    bool readyPending = false;
};

struct GamgHierarchy {
    std::vector<GamgLevel> levels;
    int nCellsInCoarsestLevel = -1, mergeLevels = -1, agglomerator = -1;
    uint64_t coeffEpoch = ~0ull;
    double* d_Apsi = nullptr;
    double* d_finestCorr = nullptr;
    double* d_finestRes = nullptr;
    // distributed coarsest-level solve in one kernel per rank (ldu_coarsest.hip, peer-store backend): -1 = not decided yet
    // (decided collectively: an and-reduce of every rank's own eligibility), 0 / 1
    int coarsestPeer = -1;
    int coarsestPeerEpoch = -1;
    int peerWgEpoch = -1;                // carrier epoch the levels' one-launch smoother decisions (ldu_addr::peerWg) belong to          // ctx->commEpoch the decision belongs to (ldu_ctx_comm_select changes the carriers)
    int* d_cycPair = nullptr;
    bool aggPending = false;             // level coefficients still being built on ctx->stream3: the main stream joins before it reads them
    hipEvent_t evFinest = nullptr;       // the finest matrix's engine copy is being filled on ctx->stream3 (refresh_level_coeffs)
    bool finestPending = false;
};

template <class T>
static int up(T** dst, const std::vector<T>& src)
{
    size_t n = src.size() ? src.size() : 1;
    LDU_CHECK_HIP(hipMalloc((void**)dst, n * sizeof(T)));
    if (src.size())
        LDU_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

void gamg_wait_plans(GamgHierarchy* g)
{
    if (!g) return;
    for (auto& L : g->levels) if (L.addr) addr_bg_wait(L.addr);
}

void gamg_free(GamgHierarchy* g)
{
    if (!g) return;
    gamg_wait_plans(g);
    if (g->aggPending && !g->levels.empty() && g->levels[0].addr && g->levels[0].addr->ctx->stream3)
        (void)hipStreamSynchronize(g->levels[0].addr->ctx->stream3);   // the level matrices are still being written
    for (auto& L : g->levels)
    {
        void* ptrs[] = {L.d_childStart, L.d_child, L.d_mapNew, L.d_cfStart, L.d_cfFine, L.d_cfFlip,
                        L.d_ccStart, L.d_ccFine, L.d_childStartO, L.d_childO, L.d_corr, L.d_src,
                        L.d_pcStart, L.d_pcFine};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        if (L.mat) matrix_free(L.mat);
        if (L.addr) { plan_free(L.addr); delete L.addr; }
        if (L.evReady) (void)hipEventDestroy(L.evReady);
    }
    if (g->d_Apsi) (void)hipFree(g->d_Apsi);
    if (g->d_finestCorr) (void)hipFree(g->d_finestCorr);
    if (g->d_finestRes) (void)hipFree(g->d_finestRes);
    if (g->d_cycPair) (void)hipFree(g->d_cycPair);
    if (g->evFinest) (void)hipEventDestroy(g->evFinest);
    delete g;
}

void gamg_invalidate_factors(GamgHierarchy* g)
{
    for (auto& L : g->levels)
    {
        if (L.mat) { L.mat->rDKind = -1; coupled_invalidate(L.mat); }
        if (L.addr && L.addr->peer) L.addr->peer->pending = false;   // (fallback_prepare: an exchange left half-way)
    }
}

// ---------------------------------------------------------------- pair agglomeration (host)

// One pairing pass (pairGAMGAgglomerate.C:31-198).  cellFaces order = faces where the cell is
// the upper (neighbour) side first, then owned faces, both ascending (:70-88) - i.e. exactly the
// losort / ownerStart row order.
static void pair_level(int nCells, int nFaces, const int* lower, const int* upper,
                       const std::vector<double>& weight, std::vector<int>& coarseMap, int& nCoarse)
{
    std::vector<int> nbrStart(nCells + 1, 0), ownStart(nCells + 1, 0), nbrFaces(nFaces);
    for (int f = 0; f < nFaces; f++) { nbrStart[upper[f] + 1]++; ownStart[lower[f] + 1]++; }
    for (int c = 0; c < nCells; c++) { nbrStart[c + 1] += nbrStart[c]; ownStart[c + 1] += ownStart[c]; }
    {
        std::vector<int> pos(nbrStart.begin(), nbrStart.end() - 1);
        for (int f = 0; f < nFaces; f++) nbrFaces[pos[upper[f]]++] = f;
    }
    coarseMap.assign(nCells, -1);
    nCoarse = 0;
    const double negGreat = -1e20;
    for (int c = 0; c < nCells; c++)
    {
        if (coarseMap[c] >= 0) continue;
        int match = -1;
        double best = negGreat;
        auto tryFace = [&](int f) {
            if (coarseMap[upper[f]] < 0 && coarseMap[lower[f]] < 0 && weight[f] > best)
            {
                match = f;
                best = weight[f];
            }
        };
        for (int t = nbrStart[c]; t < nbrStart[c + 1]; t++) tryFace(nbrFaces[t]);
        for (int f = ownStart[c]; f < ownStart[c + 1]; f++) tryFace(f);
        if (match >= 0)
        {
            coarseMap[upper[match]] = nCoarse;
            coarseMap[lower[match]] = nCoarse;
            nCoarse++;
            continue;
        }
        // no free neighbour: join the neighbouring cluster across the heaviest face (:138-170)
        int cm = -1;
        double cbest = negGreat;
        auto tryCluster = [&](int f) {
            if (weight[f] > cbest) { cm = f; cbest = weight[f]; }
        };
        for (int t = nbrStart[c]; t < nbrStart[c + 1]; t++) tryCluster(nbrFaces[t]);
        for (int f = ownStart[c]; f < ownStart[c + 1]; f++) tryCluster(f);
        if (cm >= 0) coarseMap[c] = std::max(coarseMap[upper[cm]], coarseMap[lower[cm]]);
    }
    for (int c = 0; c < nCells; c++)
        if (coarseMap[c] < 0) coarseMap[c] = nCoarse++;   // singletons (:177-184)
    for (int c = 0; c < nCells; c++) coarseMap[c] = nCoarse - 1 - coarseMap[c];   // reversal (:186-195)
}

// Coarse addressing (GAMGAgglomerateLduAddressing.C:91-198): coarse faces de-duplicated per
// (owner, neighbour) pair, numbered owner by owner, inside an owner in the order in which the fine faces (ascending)
// discover its neighbours; and restrictFaceField of the face weights (GAMGAgglomerationTemplates.C:63-83: the fine
// weights added to their coarse face in ascending fine-face order).
// Host threads, same result as the reference's sequential loop: a coarse face belongs to ONE coarse owner, so the owners are
// cut into contiguous ranges of equal face counts and every thread walks ALL fine faces in order but only handles those
// whose coarse owner lies in its range - the discovery order inside an owner and the order of the weight sums are the
// sequential ones.  (12.7 M-cell motorBike mesh: 0.55 s of every 0.81 s pairing step were this function, one thread.)
static void coarse_addressing(int nFineFaces, const int* lower, const int* upper,
                              const std::vector<int>& rmap, int nCoarse, std::vector<int>& faceRestrict,
                              std::vector<int>& cLower, std::vector<int>& cUpper, const std::vector<double>& w,
                              std::vector<double>& cw)
{
    const int nT = nFineFaces >= 400000 ? (int)std::min(12u, std::max(1u, std::thread::hardware_concurrency())) : 1;
    auto par = [&](long n, const std::function<void(int, long, long)>& fn) {
        if (nT == 1) { fn(0, 0, n); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < nT; t++) th.emplace_back(fn, t, n * t / nT, n * (t + 1) / nT);
        for (auto& t : th) t.join();
    };
    faceRestrict.assign(nFineFaces, 0);
    // coarse owner of every fine face (-1: both cells in one coarse cell) and the faces per owner (an upper bound of its
    // neighbour count)
    std::vector<int> ownOf(nFineFaces), neiOf(nFineFaces);
    std::vector<int> cap(nCoarse + 1, 0);
    par(nFineFaces, [&](int, long f0, long f1) {
        for (long f = f0; f < f1; f++)
        {
            const int a = rmap[upper[f]], b = rmap[lower[f]];
            if (a == b) { faceRestrict[f] = -(a + 1); ownOf[f] = -1; neiOf[f] = -1; continue; }
            ownOf[f] = std::min(a, b);
            neiOf[f] = std::max(a, b);
            __atomic_fetch_add(&cap[ownOf[f] + 1], 1, __ATOMIC_RELAXED);
        }
    });
    for (int c = 0; c < nCoarse; c++) cap[c + 1] += cap[c];
    const long nSlots = cap[nCoarse];
    std::vector<int> nbrOf(nSlots), cnt(nCoarse, 0), local(nFineFaces);
    std::vector<double> wSlot(nSlots, 0.0);
    // owner ranges of equal slot counts
    std::vector<int> cut(nT + 1, nCoarse);
    cut[0] = 0;
    for (int t = 1; t < nT; t++)
        cut[t] = (int)(std::lower_bound(cap.begin(), cap.end(), (int)(nSlots * t / nT)) - cap.begin());
    for (int t = 1; t <= nT; t++) cut[t] = std::max(cut[t], cut[t - 1]);
    cut[nT] = nCoarse;
    {
        auto job = [&](int t) {
            const int c0 = cut[t], c1 = cut[t + 1];
            if (c0 >= c1) return;
            for (int f = 0; f < nFineFaces; f++)
            {
                const int own = ownOf[f];
                if (own < c0 || own >= c1) continue;
                const int nei = neiOf[f];
                const int base = cap[own];
                int found = -1;
                for (int i = 0; i < cnt[own]; i++)
                    if (nbrOf[base + i] == nei) { found = i; break; }
                if (found < 0) { found = cnt[own]++; nbrOf[base + found] = nei; }
                local[f] = found;
                wSlot[base + found] += w[f];
            }
        };
        if (nT == 1) job(0);
        else
        {
            std::vector<std::thread> th;
            for (int t = 0; t < nT; t++) th.emplace_back(job, t);
            for (auto& t : th) t.join();
        }
    }
    std::vector<int> start(nCoarse + 1, 0);
    for (int c = 0; c < nCoarse; c++) start[c + 1] = start[c] + cnt[c];
    const int nCF = start[nCoarse];
    cLower.resize(nCF);
    cUpper.resize(nCF);
    cw.assign(nCF, 0.0);
    par(nCoarse, [&](int, long c0, long c1) {
        for (long c = c0; c < c1; c++)
            for (int i = 0; i < cnt[c]; i++)
            {
                const int k = start[c] + i;
                cLower[k] = (int)c;
                cUpper[k] = nbrOf[cap[c] + i];
                cw[k] = wSlot[cap[c] + i];
            }
    });
    par(nFineFaces, [&](int, long f0, long f1) {
        for (long f = f0; f < f1; f++)
            if (ownOf[f] >= 0) faceRestrict[f] = start[ownOf[f]] + local[f];
    });
}

// coarse image of one processor patch (processorGAMGInterface.C:47-126)
struct HostPatch {
    int nbrRank = -1;
    int nbrPatch = -1;            // cyclic
    std::vector<int> faceCells;   // coarse cell per coarse patch face
    std::vector<int> fra;         // faceRestrictAddressing: fine patch face -> coarse patch face
};

struct HostLevel {
    int nCells = 0;
    std::vector<int> restrictAddr, faceRestrictAddr, lower, upper;
    std::vector<HostPatch> patches;
};

// pairGAMGAgglomerate.C:201-292.  Every rank runs this loop in lock-step: the stop criterion is
// and-reduced over the ranks and the restrict maps are exchanged across the processor patches
// at every pair level (GAMGAgglomerateLduAddressing.C:201-268).
// levelReady(i): out[i] is final (no later pair level merges into it) - the caller starts that level's plan while the next
// levels are still being paired (out never reallocates: capacity is reserved up front)
static int agglomerate_all(const ldu_addr* fine, const std::vector<double>& fineWeights,
                           int nCellsInCoarsestLevel, int mergeLevels, std::vector<HostLevel>& out,
                           const std::function<void(size_t)>& levelReady)
{
    ldu_ctx* ctx = fine->ctx;
    out.clear();
    out.reserve(kMaxLevels);
    size_t announced = 0;
    std::vector<double> w = fineWeights;
    int nPairLevels = 0;
    std::vector<int> sub = fine->subOf;      // sub-domain of every cell of the level being paired (sub-domain mode)
    while ((int)out.size() < kMaxLevels - 1)
    {
        const bool top = out.empty();
        const int nC = top ? fine->nCells : out.back().nCells;
        const std::vector<int>& lo = top ? fine->l : out.back().lower;
        const std::vector<int>& up_ = top ? fine->u : out.back().upper;
        const int nF = (int)lo.size();
        HostLevel L;
        const auto tp0 = std::chrono::steady_clock::now();
        pair_level(nC, nF, lo.data(), up_.data(), w, L.restrictAddr, L.nCells);
        const auto tp1 = std::chrono::steady_clock::now();
        // continueAgglomerating (GAMGAgglomeration.C:53-62): and-reduce over the ranks
        int cont = (L.nCells >= nCellsInCoarsestLevel) ? 1 : 0;
        if (fine->nSub > 0)
        {
            // sub-domain mode: every sub-domain is a rank of the reference's run - the and-reduce happens right here
            std::vector<int> csub(L.nCells, 0), cnt(fine->nSub, 0);
            for (int c = 0; c < nC; c++) csub[L.restrictAddr[c]] = sub[c];
            for (int c = 0; c < L.nCells; c++) cnt[csub[c]]++;
            for (int d = 0; d < fine->nSub; d++) if (cnt[d] < nCellsInCoarsestLevel) cont = 0;
            sub.swap(csub);
        }
        if (comm_allreduce_min_int(ctx, &cont)) return -1;
        if (!cont) break;
        std::vector<double> cw;
        coarse_addressing(nF, lo.data(), up_.data(), L.restrictAddr, L.nCells, L.faceRestrictAddr,
                          L.lower, L.upper, w, cw);
        w.swap(cw);
        if (getenv("LDU_VERBOSE") && nC >= 1000000)
            fprintf(stderr, "[ldugpu] pairing of %d cells: pairs %.3f s, coarse addressing and weights %.3f s\n", nC,
                    std::chrono::duration<double>(tp1 - tp0).count(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - tp1).count());

        // coarse processor interfaces: coarse faces = unique (master cell, slave cell) pairs in
        // order of first occurrence - both ranks scan the same faces in the same order
        const size_t nP = fine->patches.size();
        if (nP)
        {
            std::vector<std::vector<int>> send(nP), recv;
            std::vector<const std::vector<int>*> fineFC(nP);
            for (size_t p = 0; p < nP; p++)
            {
                fineFC[p] = top ? &fine->patches[p].faceCells : &out.back().patches[p].faceCells;
                send[p].resize(fineFC[p]->size());
                for (size_t i = 0; i < send[p].size(); i++) send[p][i] = L.restrictAddr[(*fineFC[p])[i]];
            }
            if (comm_exchange_ints(ctx, fine->patches, send, recv)) return -1;
            L.patches.resize(nP);
            for (size_t p = 0; p < nP; p++)
            {
                HostPatch& HP = L.patches[p];
                HP.nbrRank = fine->patches[p].nbrRank;
                HP.nbrPatch = fine->patches[p].nbrPatch;
                // processorGAMGInterface.C:84 myProcNo() < neighbProcNo(); cyclicGAMGInterface.C:86 owner()
                // (= index < neighbour index, cyclicPolyPatch)
                const bool master = HP.nbrPatch >= 0 ? (int)p < HP.nbrPatch : ctx->rank < HP.nbrRank;
                std::map<std::pair<int, int>, int> seen;
                HP.fra.resize(send[p].size());
                for (size_t ffi = 0; ffi < send[p].size(); ffi++)
                {
                    const int loc = send[p][ffi], nbr = recv[p][ffi];
                    const std::pair<int, int> key = master ? std::make_pair(loc, nbr) : std::make_pair(nbr, loc);
                    auto it = seen.find(key);
                    if (it == seen.end())
                    {
                        it = seen.emplace(key, (int)HP.faceCells.size()).first;
                        HP.faceCells.push_back(loc);
                    }
                    HP.fra[ffi] = it->second;
                }
            }
        }

        if (nPairLevels % mergeLevels)
        {
            // combineLevels (pairGAMGAgglomerationCombineLevels.C:32-95, GAMGInterface.C:36-49)
            HostLevel& P = out.back();
            for (size_t i = 0; i < P.faceRestrictAddr.size(); i++)
            {
                int v = P.faceRestrictAddr[i];
                P.faceRestrictAddr[i] = v >= 0 ? L.faceRestrictAddr[v] : -L.restrictAddr[-v - 1] - 1;
            }
            for (size_t i = 0; i < P.restrictAddr.size(); i++) P.restrictAddr[i] = L.restrictAddr[P.restrictAddr[i]];
            P.nCells = L.nCells;
            P.lower.swap(L.lower);
            P.upper.swap(L.upper);
            for (size_t p = 0; p < P.patches.size(); p++)
            {
                for (auto& v : P.patches[p].fra) v = L.patches[p].fra[v];
                P.patches[p].faceCells.swap(L.patches[p].faceCells);
            }
        }
        else
        {
            // a new level begins: everything before it is final
            for (; announced < out.size(); announced++) levelReady(announced);
            out.push_back(std::move(L));
            if (mergeLevels == 1) { levelReady(announced); announced++; }
        }
        nPairLevels++;
    }
    for (; announced < out.size(); announced++) levelReady(announced);
    return 0;
}

// ---------------------------------------------------------------- hierarchy construction

// restriction / prolongation maps (cells) and coefficient-agglomeration lists (faces) of one level: the two halves are
// independent and run side by side on large levels.  Needs this level's and the finer level's plan (perm / iperm).
static int build_level_maps(GamgLevel& L, const ldu_addr* cA, const ldu_addr* fineA, const std::vector<int>& restrictAddr,
                            const std::vector<int>& faceRestrictAddr)
{
    const int nFC = fineA->nCells, nFF = fineA->nFaces, nCC = cA->nCells, nCF = cA->nFaces;
    const int dev = cA->ctx->device;
    int rcCells = 0, rcFaces = 0;
    std::string errCells, errFaces;
    auto cells = [&]() {
        if (hipSetDevice(dev) != hipSuccess) { rcCells = -1; errCells = "hipSetDevice failed"; return; }
        // children per coarse cell, ascending original fine index
        std::vector<int> childStartO(nCC + 1, 0), childO(nFC);
        for (int i = 0; i < nFC; i++) childStartO[restrictAddr[i] + 1]++;
        for (int c = 0; c < nCC; c++) childStartO[c + 1] += childStartO[c];
        {
            std::vector<int> pos(childStartO.begin(), childStartO.end() - 1);
            for (int i = 0; i < nFC; i++) childO[pos[restrictAddr[i]]++] = i;
        }
        // the same in NEW numbering of both levels (order of summation unchanged)
        std::vector<int> childStart(nCC + 1, 0), child(nFC), mapNew(nFC);
        for (int cn = 0; cn < nCC; cn++)
        {
            int co = cA->perm[cn];
            childStart[cn + 1] = childStart[cn] + (childStartO[co + 1] - childStartO[co]);
        }
        for (int cn = 0; cn < nCC; cn++)
        {
            int co = cA->perm[cn];
            int k = childStart[cn];
            for (int t = childStartO[co]; t < childStartO[co + 1]; t++) child[k++] = fineA->iperm[childO[t]];
        }
        for (int fn = 0; fn < nFC; fn++) mapNew[fn] = cA->iperm[restrictAddr[fineA->perm[fn]]];
        if (up(&L.d_childStart, childStart) || up(&L.d_child, child) || up(&L.d_mapNew, mapNew)
            || up(&L.d_childStartO, childStartO) || up(&L.d_childO, childO)
            || hipMalloc((void**)&L.d_corr, sizeof(double) * (size_t)(nCC + 1)) != hipSuccess
            || hipMalloc((void**)&L.d_src, sizeof(double) * (size_t)(nCC + 1)) != hipSuccess)
        {
            rcCells = -1;
            errCells = ldu_last_error_string();
            if (errCells.empty()) errCells = "device allocation of the level vectors failed";
        }
    };
    auto faces = [&]() {
        // coefficient agglomeration lists (ascending fine face): GAMGSolverAgglomerateMatrix.C:148-205
        std::vector<int> cfStart(nCF + 1, 0), cfFine, ccStart(nCC + 1, 0), ccFine;
        std::vector<unsigned char> cfFlip;
        for (int f = 0; f < nFF; f++)
        {
            int v = faceRestrictAddr[f];
            if (v >= 0) cfStart[v + 1]++;
            else ccStart[-1 - v + 1]++;
        }
        for (int c = 0; c < nCF; c++) cfStart[c + 1] += cfStart[c];
        for (int c = 0; c < nCC; c++) ccStart[c + 1] += ccStart[c];
        cfFine.resize(cfStart[nCF]);
        cfFlip.resize(cfStart[nCF] ? cfStart[nCF] : 1);
        ccFine.resize(ccStart[nCC]);
        {
            std::vector<int> p1(cfStart.begin(), cfStart.end() - 1), p2(ccStart.begin(), ccStart.end() - 1);
            for (int f = 0; f < nFF; f++)
            {
                int v = faceRestrictAddr[f];
                if (v >= 0)
                {
                    int k = p1[v]++;
                    cfFine[k] = f;
                    // orientation test against restrictAddr[l[f]] (:158-171)
                    int rl = restrictAddr[fineA->l[f]];
                    if (cA->l[v] == rl) cfFlip[k] = 0;
                    else if (cA->u[v] == rl) cfFlip[k] = 1;
                    else { rcFaces = -5; errFaces = "GAMG: inconsistent addressing between fine and coarse grids"; return; }
                }
                else ccFine[p2[-1 - v]++] = f;
            }
        }
        if (up(&L.d_cfStart, cfStart) || up(&L.d_cfFine, cfFine) || up(&L.d_cfFlip, cfFlip)
            || up(&L.d_ccStart, ccStart) || up(&L.d_ccFine, ccFine))
        {
            rcFaces = -1;
            errFaces = ldu_last_error_string();
        }
    };
    if (nFC >= 200000)
    {
        std::thread t(cells);
        faces();
        t.join();
    }
    else { cells(); faces(); }
    if (rcCells) { ldu_set_error(errCells); return rcCells; }
    if (rcFaces) { ldu_set_error(errFaces); return rcFaces; }
    return 0;
}

static int refresh_level_coeffs(ldu_matrix* m, const ldu_controls* c);
// deferCoeffs: the caller runs refresh_level_coeffs itself (gamg_solve: behind the launches of its initial residual - the chain
// is ~150 launches, 1.2-1.4 ms of HOST time during which the main stream used to sit idle)
static int ensure_hierarchy(ldu_matrix* m, const ldu_controls* c, bool deferCoeffs = false)
{
    ldu_addr* a = m->a;
    GamgHierarchy* g = m->gamg;
    const bool reuse = g && c->cacheAgglomeration && g->nCellsInCoarsestLevel == c->nCellsInCoarsestLevel
                       && g->mergeLevels == c->mergeLevels && g->agglomerator == c->agglomerator
                       && c->agglomerator == LDU_AGG_FACEAREAPAIR;
    if (!reuse)
    {
        if (g) { gamg_free(g); m->gamg = nullptr; }
        g = new GamgHierarchy();
        // whatever goes wrong below: the half-built hierarchy is released (every error path just returns)
        struct Guard { GamgHierarchy* g; ~Guard() { if (g) gamg_free(g); } } guard{g};
        g->nCellsInCoarsestLevel = c->nCellsInCoarsestLevel;
        g->mergeLevels = c->mergeLevels;
        g->agglomerator = c->agglomerator;
        std::vector<double> w(a->nFaces);
        if (c->agglomerator == LDU_AGG_ALGEBRAICPAIR)
        {
            // algebraicPairGAMGAgglomeration.C:55: mag(matrix.upper())
            if (a->nFaces)
                LDU_CHECK_HIP(hipMemcpy(w.data(), m->d_upperO, sizeof(double) * a->nFaces, hipMemcpyDeviceToHost));
            for (auto& x : w) x = std::fabs(x);
        }
        else
        {
            if ((int)a->faceWeights.size() != a->nFaces)
            {
                ldu_set_error("faceAreaPair agglomeration needs ldu_addr_set_face_weights()");
                return -7;
            }
            w = a->faceWeights;
        }
        const bool verbose = getenv("LDU_VERBOSE") != nullptr;
        const auto tSetup0 = std::chrono::steady_clock::now();
        auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - tSetup0).count(); };
        std::vector<HostLevel> hl;
        const bool parallelPlans = !getenv("LDU_NO_PARALLEL_PLANS");
        // The level plans (dependency levels, sliced-ELL tables) are independent of each other and of the pairing of the
        // levels below them: each starts on its own host thread the moment its level's addressing is final, while
        // agglomerate_all goes on pairing the next level (round 3 paired all levels first: 0.7 s of the 216^3 set-up
        // with nothing else running).
        g->levels.resize(kMaxLevels);
        std::vector<int> rcs(kMaxLevels, 0);
        std::vector<std::string> errs(kMaxLevels);
        std::vector<std::thread> th;
        const bool prebuild = parallelPlans && !getenv("LDU_NO_CLUSTER_PREBUILD");
        int rcFinest = 0;
        std::string errFinest;
        // cluster plans run on threads of their own (joined at the very end: nothing below needs them), so that the level
        // maps can be built as soon as the level plans are there
        std::vector<std::thread> cth;
        std::mutex cthMu;
        std::vector<int> crc(kMaxLevels, 0);
        std::vector<std::string> cerr(kMaxLevels);
        struct JoinAll { std::vector<std::thread>& v; ~JoinAll() { for (auto& t : v) if (t.joinable()) t.join(); } } joinClusters{cth};
        // the task orders of the pipelined GaussSeidel sweeps a level will be asked for (lazily built at the first smoothing
        // call otherwise: 2.3 s of the first solve on the 12.7 M-cell motorBike mesh, one level after the other)
        // Large levels: on a thread of their own that OUTLIVES this set-up (ldu_addr::bgPlan) - nothing needs these plans for
        // correctness, the level engines sweep until they are there (first solve on the 12.7 M-cell mesh 3.9 -> 2.5 s of host
        // time before the first V-cycle; ldu_matrix_wait_plans / the engine queries wait for them).  LDU_BG_PLANS=0: as before.
        static const bool bgPlans = !getenv("LDU_BG_PLANS") || atoi(getenv("LDU_BG_PLANS")) != 0;
        const int smootherKind = c->smoother;
        const int devId = a->ctx->device;
        auto sweepPlans = [&, smootherKind, devId](ldu_addr* A, int nPost, int nPre) {
            if (smootherKind != LDU_SM_GAUSSSEIDEL && smootherKind != LDU_SM_NONBLOCKINGGAUSSSEIDEL) return;
            auto build = [A, nPost, nPre]() {
                for (int n : {nPost, nPre})
                    if (n >= 2) (void)k_gs_prebuild(A, n > 4 ? 4 : n);   // (errors: the smoothing call builds again and says)
            };
            addr_bg_wait(A);      // (a hierarchy rebuilt for other controls: the previous plan thread of this addressing first)
            if (bgPlans && A->nCells >= 50000 && !A->nPatchFaces)
            {
                A->bgState.store(1, std::memory_order_release);
                A->bgPlan = std::thread([A, build, devId]() {
                    tl_bgPlanThread = true;
                    if (hipSetDevice(devId) == hipSuccess) build();
                    A->bgState.store(2, std::memory_order_release);
                });
            }
            else build();
        };
        if (prebuild)   // the finest level's cluster plan (the longest single piece, 1.4 s at 216^3) beside everything else
            cth.emplace_back([&]() {
                if (hipSetDevice(a->ctx->device) != hipSuccess) { rcFinest = -1; errFinest = "hipSetDevice failed"; return; }
                rcFinest = k_cluster_build_one(a);
                if (rcFinest) errFinest = ldu_last_error_string();
                else sweepPlans(a, c->nFinestSweeps, 0);
            });
        // per level: 0 = plan pending, 1 = plan there, -1 = failed.  The maps of level i (restriction / prolongation /
        // coefficient agglomeration) need the plans of levels i and i - 1 only: they start on level i's thread as soon as
        // both are there, not after the last level's plan (216^3: level 1's maps, 0.4 s, used to start at 1.07 s)
        std::vector<int> planState(kMaxLevels, 0);
        std::vector<int> mrc(kMaxLevels, 0);
        std::vector<std::string> merr(kMaxLevels);
        std::mutex planMu;
        std::condition_variable planCv;
        auto levelReady = [&](size_t i) {
            if (!parallelPlans) return;
            const double tAnnounced = since();
            th.emplace_back([&, i, tAnnounced]() {
                auto publish = [&](int v) {
                    { std::lock_guard<std::mutex> lk(planMu); planState[i] = v; }
                    planCv.notify_all();
                };
                if (hipSetDevice(a->ctx->device) != hipSuccess) { rcs[i] = -1; errs[i] = "hipSetDevice failed"; publish(-1); return; }
                rcs[i] = addr_create_internal(a->ctx, &g->levels[i].addr, hl[i].nCells, (int)hl[i].lower.size(),
                                              hl[i].lower.data(), hl[i].upper.data());
                if (rcs[i]) { errs[i] = ldu_last_error_string(); publish(-1); return; }
                publish(1);
                const double tPlanDone = since();
                // the cluster plan of the level right behind its level plan, on a thread of its own
                if (prebuild && hl[i].patches.empty())
                {
                    std::lock_guard<std::mutex> lk(cthMu);
                    cth.emplace_back([&, i]() {
                        if (hipSetDevice(a->ctx->device) != hipSuccess) { crc[i] = -1; cerr[i] = "hipSetDevice failed"; return; }
                        crc[i] = k_cluster_build_one(g->levels[i].addr);
                        if (crc[i]) cerr[i] = ldu_last_error_string();
                        else
                            sweepPlans(g->levels[i].addr, std::min(c->nPostSweeps + c->postSweepsLevelMultiplier * (int)i, c->maxPostSweeps),
                                       c->nPreSweeps ? std::min(c->nPreSweeps + c->preSweepsLevelMultiplier * (int)i, c->maxPreSweeps) : 0);
                    });
                }
                const ldu_addr* fa = a;
                if (i > 0)
                {
                    std::unique_lock<std::mutex> lk(planMu);
                    planCv.wait(lk, [&]() { return planState[i - 1] != 0; });
                    if (planState[i - 1] < 0) return;      // (reported by that level)
                    fa = g->levels[i - 1].addr;
                }
                const double tMapsStart = since();
                mrc[i] = build_level_maps(g->levels[i], g->levels[i].addr, fa, hl[i].restrictAddr, hl[i].faceRestrictAddr);
                if (mrc[i]) merr[i] = ldu_last_error_string();
                if (verbose && hl[i].nCells >= 100000)
                    fprintf(stderr, "[ldugpu] set-up of level %zu (%d cells): announced at %.3f s, plan done at %.3f s, maps %.3f - %.3f s\n",
                            i + 1, hl[i].nCells, tAnnounced, tPlanDone, tMapsStart, since());
            });
        };
        const int rcAgg = agglomerate_all(a, w, c->nCellsInCoarsestLevel, std::max(1, c->mergeLevels), hl, levelReady);
        const double tAgg = since();
        for (auto& t : th) t.join();
        g->levels.resize(hl.size());
        if (rcAgg) return -1;
        if (hl.empty())
        {
            // GAMGSolver.C:108-126
            ldu_set_error("No coarse levels created, either matrix too small for GAMG or "
                          "nCellsInCoarsestLevel too large.");
            return -8;
        }
        for (size_t i = 0; i < hl.size(); i++)
            if (rcs[i]) { ldu_set_error("GAMG level plan: " + errs[i]); return -1; }
        const double tPlans = since();
        const ldu_addr* fineA = a;
        for (size_t i = 0; i < hl.size(); i++)
        {
            GamgLevel& L = g->levels[i];
            L.nFineCells = fineA->nCells;
            L.nFineFaces = fineA->nFaces;
            L.restrictAddr.swap(hl[i].restrictAddr);
            L.faceRestrictAddr.swap(hl[i].faceRestrictAddr);
            if (!parallelPlans
                && addr_create_internal(a->ctx, &L.addr, hl[i].nCells, (int)hl[i].lower.size(),
                                        hl[i].lower.data(), hl[i].upper.data()))
                return -1;
            // coarse processor patches + the lists that agglomerate their coefficients
            // (GAMGInterface::agglomerateCoeffs, GAMGInterface.C:61-75) in ascending fine-face order
            if (!hl[i].patches.empty())
            {
                for (auto& hp : hl[i].patches)
                {
                    Patch P;
                    P.n = (int)hp.faceCells.size();
                    P.nbrRank = hp.nbrRank;
                    P.nbrPatch = hp.nbrPatch;
                    P.faceCells = hp.faceCells;
                    L.addr->patches.push_back(P);
                }
                if (plan_finalize_patches(L.addr)) return -1;
                std::vector<int> pcStart(L.addr->nPatchFaces + 1, 0), pcFine(fineA->nPatchFaces);
                for (size_t p = 0; p < hl[i].patches.size(); p++)
                    for (int v : hl[i].patches[p].fra) pcStart[L.addr->patches[p].offset + v + 1]++;
                for (int k = 0; k < L.addr->nPatchFaces; k++) pcStart[k + 1] += pcStart[k];
                std::vector<int> pos(pcStart.begin(), pcStart.end() - 1);
                for (size_t p = 0; p < hl[i].patches.size(); p++)
                    for (size_t ffi = 0; ffi < hl[i].patches[p].fra.size(); ffi++)
                        pcFine[pos[L.addr->patches[p].offset + hl[i].patches[p].fra[ffi]]++] =
                            fineA->patches[p].offset + (int)ffi;
                if (up(&L.d_pcStart, pcStart) || up(&L.d_pcFine, pcFine)) return -1;
            }
            if (matrix_alloc(L.addr, &L.mat)) return -1;
            if (getenv("LDU_VERBOSE"))
                fprintf(stderr, "[ldugpu] GAMG level %2zu: %9d cells %9d faces  %5d dependency levels  %7d slices\n",
                        i + 1, L.addr->nCells, L.addr->nFaces, L.addr->nLevels, L.addr->nSlices);
            fineA = L.addr;
        }
        if (!parallelPlans)
            for (size_t i = 0; i < hl.size(); i++)
                if (build_level_maps(g->levels[i], g->levels[i].addr, i == 0 ? a : g->levels[i - 1].addr, g->levels[i].restrictAddr,
                                     g->levels[i].faceRestrictAddr))
                    return -1;
        for (size_t i = 0; i < hl.size(); i++)
            if (mrc[i]) { ldu_set_error("GAMG level maps: " + merr[i]); return -1; }
        const double tMaps = since();
        {
            std::lock_guard<std::mutex> lk(cthMu);     // (no plan thread is alive any more: the list is complete)
            for (auto& t : cth) t.join();
        }
        for (size_t i = 0; i < hl.size(); i++)
            if (crc[i]) { ldu_set_error("cluster plan: " + cerr[i]); return -1; }
        if (rcFinest) { ldu_set_error("cluster plan: " + errFinest); return -1; }
        if (!getenv("LDU_NO_CLUSTER_PREBUILD"))
        {
            // the cluster plans of the large levels (and of the finest matrix), one host thread each
            std::vector<ldu_addr*> big{a};
            for (auto& L : g->levels) big.push_back(L.addr);
            if (k_cluster_prebuild(big)) return -1;
        }
        if (verbose)
            fprintf(stderr, "[ldugpu] GAMG set-up: pairing %.3f s, level plans done at %.3f s, level maps at %.3f s, cluster plans "
                            "at %.3f s (%zu levels)\n", tAgg, tPlans, tMaps, since(), hl.size());
        const size_t n = (size_t)a->nCells + 1;
        LDU_CHECK_HIP(hipMalloc((void**)&g->d_Apsi, sizeof(double) * n));
        LDU_CHECK_HIP(hipMalloc((void**)&g->d_finestCorr, sizeof(double) * n));
        LDU_CHECK_HIP(hipMalloc((void**)&g->d_finestRes, sizeof(double) * n));
        guard.g = nullptr;
        m->gamg = g;
    }
    return deferCoeffs ? 0 : refresh_level_coeffs(m, c);
}

// level coefficients: rebuilt from the fine matrix whenever the coefficients changed
// (GAMGSolver.C:86-89 runs agglomerateMatrix for every level in every solver construction)
static int refresh_level_coeffs(ldu_matrix* m, const ldu_controls* c)
{
    ldu_addr* a = m->a;
    GamgHierarchy* g = m->gamg;
    if (g->coeffEpoch != m->coeffEpoch)
    {
        // On a stream of its own, beside the finest level's own work of this solve (its level layout, the initial residual,
        // the normalisation factor, the restrictions): the chain below - two gathers, a permutation and a layout fill per
        // level, every level from the one above it - is a few large launches and ~80 small ones.  It starts when the finest
        // LDU arrays are in place (event of ldu_matrix_set_coeffs; otherwise: now) and vcycle() joins before the first level
        // matrix is read.
        ldu_ctx* ctx = a->ctx;
        hipStream_t s = ctx->stream;
        if (ctx->aggOverlap && ctx->stream3)
        {
            if (g->aggPending) LDU_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->evAggJoin, 0));   // (an unfinished earlier chain)
            if (!(ctx->aggForkOf == m && ctx->aggForkEpoch == m->coeffEpoch))
                LDU_CHECK_HIP(hipEventRecord(ctx->evAggFork, ctx->stream));
            LDU_CHECK_HIP(hipStreamWaitEvent(ctx->stream3, ctx->evAggFork, 0));
            s = ctx->stream3;
        }
        // 1. the face-ordered coefficients of every level, each from the level above it (the only sequential part)
        const ldu_matrix* fm = m;
        for (auto& L : g->levels)
        {
            ldu_matrix* cm = L.mat;
            cm->sym = m->sym;
            if (!cm->sym && cm->d_lowerO == cm->d_upperO)
            {
                LDU_CHECK_HIP(hipMalloc((void**)&cm->d_lowerO, sizeof(double) * (size_t)(L.addr->nFaces + 1)));
            }
            if (k_agglomerate_coeffs(L.addr->nFaces, L.d_cfStart, L.d_cfFine, L.d_cfFlip, L.addr->nCells,
                                     L.d_ccStart, L.d_ccFine, L.d_childStartO, L.d_childO, fm->d_diagO,
                                     fm->d_upperO, fm->d_lowerO, cm->d_diagO, cm->d_upperO, cm->d_lowerO,
                                     m->sym, s))
                return -1;
            if (L.addr->nPatchFaces)
                if (k_patch_agglomerate(L.addr->nPatchFaces, L.d_pcStart, L.d_pcFine, fm->d_bou, fm->d_int,
                                        cm->d_bou, cm->d_int, s))
                    return -1;
            fm = cm;
        }
        // 2. per level: the level layout and the smoothing engine's own copies of the coefficients (what the first smoothing
        //    call used to fill on the main stream), an event behind them.  In the order in which a V-cycle first reads the level
        //    matrices - without pre-smoothing from the coarsest level up (the way down is restrictions only), so that the large
        //    fills of the fine levels run beside the sweeps of the coarse ones, which leave the memory system nearly idle
        const int nLv = (int)g->levels.size();
        for (int q = 0; q < nLv; q++)
        {
            GamgLevel& L = g->levels[c->nPreSweeps ? q : nLv - 1 - q];
            if (matrix_refresh_layout(L.mat, s)) return -1;
            if (ctx->aggPrefill && dev_smooth_prefill(L.mat, c->smoother, s)) return -1;
            if (s != ctx->stream)
            {
                if (!L.evReady) LDU_CHECK_HIP(hipEventCreateWithFlags(&L.evReady, hipEventDisableTiming));
                LDU_CHECK_HIP(hipEventRecord(L.evReady, s));
                L.readyPending = true;
            }
        }
        // ... and the finest matrix's own engine copy (its level layout was filled on the main stream by ldu_matrix_set_coeffs,
        // after the fork: the side stream waits for the main stream as it is now), needed last of all
        if (s != ctx->stream && ctx->aggPrefill)
        {
            if (!g->evFinest) LDU_CHECK_HIP(hipEventCreateWithFlags(&g->evFinest, hipEventDisableTiming));
            LDU_CHECK_HIP(hipEventRecord(g->evFinest, ctx->stream));
            LDU_CHECK_HIP(hipStreamWaitEvent(s, g->evFinest, 0));
            if (dev_smooth_prefill(m, c->smoother, s)) return -1;
            LDU_CHECK_HIP(hipEventRecord(g->evFinest, s));
            g->finestPending = true;
        }
        if (s != ctx->stream)
        {
            LDU_CHECK_HIP(hipEventRecord(ctx->evAggJoin, s));
            g->aggPending = true;
        }
        g->coeffEpoch = m->coeffEpoch;
    }
    return 0;
}

// ---------------------------------------------------------------- V-cycle

// GAMGSolverScale.C:31-75
static int gamg_scale(ldu_matrix* A, double* field, double* Acf, const double* source)
{
    ldu_ctx* ctx = A->a->ctx;
    hipStream_t s = ctx->stream;
    const int n = A->a->nCells;
    if (dev_amul(A, Acf, field, false)) return -1;
    // (source.field , Acf.field) fused, one 2-double all-reduce (:54-61)
    if (k_reduce(ctx, n, RED_DOT2, source, field, Acf, nullptr, S_SCALE_NUM, s)) return -1;
    if (comm_allreduce_scalars(ctx, S_SCALE_NUM, 2, s)) return -1;
    return k_gamg_scale_update(n, field, source, Acf, A->d_diag, ctx->S(), s);
}

// GAMGSolverInterpolate.C:30-83
static int gamg_interpolate(ldu_matrix* A, double* psi, double* Apsi)
{
    hipStream_t s = A->a->ctx->stream;
    if (k_offdiag(A, Apsi, psi, 2, s)) return -1;
    return k_neg_div(A->a->nCells, psi, Apsi, A->d_diag, s);
}

// coarsest level: ICCG / BICCG with the outer tolerances (GAMGSolverSolve.C:430-487, ICCG.C:46)
static int solve_coarsest(GamgHierarchy* g, ldu_matrix* A, const ldu_controls* c, double* corr, const double* src)
{
    hipStream_t s = A->a->ctx->stream;
    if (c->directSolveCoarsest) return k_coarsest_lu(A, corr, src, g->coeffEpoch);   // GAMGSolverSolve.C:436-440
    if (k_ew(A->a->nCells, EW_ZERO, corr, nullptr, nullptr, s)) return -1;
    if (A->a->nPatchFaces || A->a->ctx->nRanks > 1)
    {
        // several ranks / coupled patches: the whole distributed Krylov solve in one kernel per rank when the peer-store
        // backend carries halos and sums and every rank's coarsest level fits (collective decision, once per hierarchy)
        if (g->coarsestPeer < 0 || g->coarsestPeerEpoch != A->a->ctx->commEpoch)
        {
            g->coarsestPeerEpoch = A->a->ctx->commEpoch;
            if (g->d_cycPair) { (void)hipFree(g->d_cycPair); g->d_cycPair = nullptr; }
            int ok = k_coarsest_peer_eligible(A) ? 1 : 0;
            if (comm_allreduce_min_int(A->a->ctx, &ok)) return -1;
            g->coarsestPeer = ok;
            if (ok)
            {
                std::vector<int> cyc(A->a->nPatchFaces, -1);
                for (auto& P : A->a->patches)
                    if (P.nbrPatch >= 0)
                        for (int f = 0; f < P.n; f++) cyc[P.offset + f] = A->a->patches[P.nbrPatch].offset + f;
                if (up(&g->d_cycPair, cyc)) return -1;
            }
            if (getenv("LDU_VERBOSE"))
                fprintf(stderr, "[ldugpu] coarsest level (%d cells, %d coupled faces): %s\n", A->a->nCells, A->a->nPatchFaces,
                        ok ? "distributed Krylov solve in one kernel per rank (peer stores)" : "host-driven Krylov loop");
        }
        if (g->coarsestPeer && A->a->ctx->sweepP2P)     // (an engine-fallback re-run takes the host-driven loop below)
        {
            const int rc = k_coarsest_solve_peer(A, c->tolerance, c->relTol, 1000, corr, src, g->d_cycPair);
            if (rc <= 0) return rc;
        }
    }
    else
    {
        // the whole Krylov solve of a tiny level in one single-wavefront kernel (ldu_coarsest.hip)
        const int rc = k_coarsest_solve(A, c->tolerance, c->relTol, 1000, corr, src);
        if (rc <= 0) return rc;
    }
    ldu_controls cc;
    ldu_default_controls(&cc);
    cc.tolerance = c->tolerance;
    cc.relTol = c->relTol;
    cc.historyCapacity = 0;
    if (A->sym) { cc.solver = LDU_SOLVER_PCG; cc.preconditioner = LDU_PRE_DIC; }
    else { cc.solver = LDU_SOLVER_PBICG; cc.preconditioner = LDU_PRE_DILU; }
    ldu_perf p;
    memset(&p, 0, sizeof(p));
    ldu_ctx* ctx = A->a->ctx;
    if (ctx->sb + 2 * S_BANK > S_NSLOTS) { ldu_set_error("scalar bank overflow"); return -1; }
    ctx->sb += S_BANK;   // nested solve: own scalar bank
    int rc = dev_solve(A, &cc, corr, src, &p, nullptr);
    ctx->sb -= S_BANK;
    return rc;
}

// Which levels smooth with gs_wg_peer_kernel (k sweeps and their boundary exchanges in one launch): a launch that talks to
// its neighbours must be taken by every rank or by none, so each level's own eligibility is and-reduced over the ranks -
// here, where every rank passes with the same list of levels, not at the first use (a rank without coupled faces on a level
// would never get there).  Repeated when the carriers of the context change (ldu_ctx_comm_select).
static int gamg_decide_peer_smoothers(ldu_matrix* m)
{
    GamgHierarchy* g = m->gamg;
    ldu_ctx* ctx = m->a->ctx;
    if (g->peerWgEpoch == ctx->commEpoch) return 0;
    g->peerWgEpoch = ctx->commEpoch;
    if (!ctx->comm) return 0;
    for (auto& L : g->levels)
    {
        ldu_addr* a = L.addr;
        int ok = k_wg_peer_eligible(a) ? 1 : 0;
        // (a rank without coupled faces on this level neither sends nor receives there: it does not veto)
        if (!a->nPatchFaces) ok = 1;
        if (comm_allreduce_min_int(ctx, &ok)) return -1;
        a->peerWg = (ok && a->nPatchFaces) ? 1 : 0;
        a->peerWgEpoch = ctx->commEpoch;
        if (a->peerWg && !a->d_cycPair)
        {
            std::vector<int> cyc(a->nPatchFaces, -1);
            for (auto& P : a->patches)
                if (P.nbrPatch >= 0)
                    for (int f = 0; f < P.n; f++) cyc[P.offset + f] = a->patches[P.nbrPatch].offset + f;
            if (up(&a->d_cycPair, cyc)) return -1;
        }
        if (getenv("LDU_VERBOSE") && a->nPatchFaces)
            fprintf(stderr, "[ldugpu] level of %d cells, %d coupled faces: GaussSeidel sweeps %s\n", a->nCells, a->nPatchFaces,
                    a->peerWg ? "and their exchanges in one launch (peer stores)" : "one by one, exchange between them");
    }
    // pipelined sweeps with remote interfaces on the block engine (ldu_blocks.hip, "Remote interfaces"): the finest level and
    // every coarse one the one-launch smoother above did not take, the same list on every rank
    if (k_blocks_peer_decide(m->a)) return -1;
    for (auto& L : g->levels) if (k_blocks_peer_decide(L.addr)) return -1;
    return 0;
}

// the level matrices are being built on ctx->stream3 (ensure_hierarchy): the main stream waits here, before it reads one
static int gamg_join_levels(GamgHierarchy* g, ldu_ctx* ctx)
{
    if (!g->aggPending) return 0;
    LDU_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->evAggJoin, 0));
    g->aggPending = false;
    g->finestPending = false;
    for (auto& L : g->levels) L.readyPending = false;
    return 0;
}

// ... or for ONE level's matrix, before its first use in a V-cycle
static int gamg_join_level(GamgHierarchy* g, ldu_ctx* ctx, int leveli)
{
    GamgLevel& L = g->levels[leveli];
    if (!L.readyPending) return 0;
    LDU_CHECK_HIP(hipStreamWaitEvent(ctx->stream, L.evReady, 0));
    L.readyPending = false;
    return 0;
}

// GAMGSolverSolve.C:120-364
static int vcycle(ldu_matrix* m, const ldu_controls* c, double* psi, const double* source, double* Apsi,
                  double* finestCorrection, double* finestResidual)
{
    GamgHierarchy* g = m->gamg;
    ldu_ctx* ctx = m->a->ctx;
    hipStream_t s = ctx->stream;
    const int coarsestLevel = (int)g->levels.size() - 1;
    const int scaleCorrection = c->scaleCorrection < 0 ? (m->sym ? 1 : 0) : c->scaleCorrection;
    auto& Lv = g->levels;
    if (gamg_decide_peer_smoothers(m)) return -1;
    // (without pre-smoothing the way down is restrictions only: the level matrices are first read at the coarsest level)
    if (c->nPreSweeps && gamg_join_levels(g, ctx)) return -1;

    if (k_restrict(Lv[0].addr->nCells, Lv[0].d_childStart, Lv[0].d_child, finestResidual, Lv[0].d_src, s))
        return -1;

    for (int leveli = 0; leveli < coarsestLevel; leveli++)
    {
        GamgLevel& L = Lv[leveli];
        const int n = L.addr->nCells;
        if (c->nPreSweeps)
        {
            if (k_ew(n, EW_ZERO, L.d_corr, nullptr, nullptr, s)) return -1;
            if (dev_smooth(L.mat, c->smoother, L.d_corr, L.d_src,
                           std::min(c->nPreSweeps + c->preSweepsLevelMultiplier * leveli, c->maxPreSweeps)))
                return -1;
            double* ACf = Apsi;
            if (scaleCorrection && leveli < coarsestLevel - 1)
                if (gamg_scale(L.mat, L.d_corr, ACf, L.d_src)) return -1;
            if (dev_amul(L.mat, ACf, L.d_corr, false)) return -1;
            if (k_ew(n, EW_SUB_INPLACE, L.d_src, ACf, nullptr, s)) return -1;
        }
        GamgLevel& C = Lv[leveli + 1];
        if (k_restrict(C.addr->nCells, C.d_childStart, C.d_child, L.d_src, C.d_src, s)) return -1;
    }

    {
        static const bool timeCoarsest = getenv("LDU_GAMG_TIME") != nullptr;
        std::chrono::steady_clock::time_point t0;
        if (timeCoarsest) { (void)hipStreamSynchronize(s); t0 = std::chrono::steady_clock::now(); }
        if (gamg_join_level(g, ctx, coarsestLevel)) return -1;
        if (solve_coarsest(g, Lv[coarsestLevel].mat, c, Lv[coarsestLevel].d_corr, Lv[coarsestLevel].d_src)) return -1;
        if (timeCoarsest)
        {
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "[ldugpu] coarsest level (%d cells): %.3f ms\n", Lv[coarsestLevel].addr->nCells,
                    1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
    }

    for (int leveli = coarsestLevel - 1; leveli >= 0; leveli--)
    {
        GamgLevel& L = Lv[leveli];
        GamgLevel& C = Lv[leveli + 1];
        const int n = L.addr->nCells;
        double* preSmoothed = finestCorrection;
        if (c->nPreSweeps)
            if (k_ew(n, EW_COPY, preSmoothed, L.d_corr, nullptr, s)) return -1;
        if (k_prolong(n, C.d_mapNew, C.d_corr, L.d_corr, s)) return -1;
        if (gamg_join_level(g, ctx, leveli)) return -1;
        double* ACf = Apsi;
        if (c->interpolateCorrection)
            if (gamg_interpolate(L.mat, L.d_corr, ACf)) return -1;
        if (scaleCorrection && leveli < coarsestLevel - 1)
            if (gamg_scale(L.mat, L.d_corr, ACf, L.d_src)) return -1;
        if (c->nPreSweeps)
            if (k_ew(n, EW_ADD_INPLACE, L.d_corr, preSmoothed, nullptr, s)) return -1;
        static const bool timeLevels = getenv("LDU_GAMG_TIME") != nullptr;   // diagnostic: per-level smoothing time
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timeLevels)
        {
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, s);
        }
        const int nPost = std::min(c->nPostSweeps + c->postSweepsLevelMultiplier * leveli, c->maxPostSweeps);
        if (dev_smooth(L.mat, c->smoother, L.d_corr, L.d_src, nPost)) return -1;
        if (timeLevels)
        {
            float ms = 0;
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            long span = 0;
            for (int i = 0; i < L.addr->nSlabs; i++) span += L.addr->slabLevelSpan[i];
            fprintf(stderr, "[ldugpu] level %2d: %9d cells %5d dag-levels %d slabs (spans %.2f x levels) width %.1f maxrow %d: %d sweeps %.3f ms\n",
                    leveli + 1, n, L.addr->nLevels, L.addr->nSlabs, (double)span / std::max(1, L.addr->nLevels),
                    L.addr->slabWidth, L.addr->maxRowWidth, nPost, ms);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
    }

    const int n0 = m->a->nCells;
    if (k_prolong(n0, Lv[0].d_mapNew, Lv[0].d_corr, finestCorrection, s)) return -1;
    if (c->interpolateCorrection)
        if (gamg_interpolate(m, finestCorrection, Apsi)) return -1;
    if (scaleCorrection)
        if (gamg_scale(m, finestCorrection, Apsi, finestResidual)) return -1;
    if (k_ew(n0, EW_ADD_INPLACE, psi, finestCorrection, nullptr, s)) return -1;
    if (g->finestPending)
    {
        LDU_CHECK_HIP(hipStreamWaitEvent(s, g->evFinest, 0));
        g->finestPending = false;
    }
    return dev_smooth(m, c->smoother, psi, source, c->nFinestSweeps);
}

static bool check_convergence(ldu_perf* p, double tol, double relTol)
{
    p->converged = (p->finalResidual < tol
                    || (relTol > kSmall && p->finalResidual < relTol * p->initialResidual)) ? 1 : 0;
    return p->converged != 0;
}

// GAMGSolverSolve.C:34-117
int gamg_solve(ldu_matrix* m, const ldu_controls* c, double* psi, const double* source, ldu_perf* perf,
               double* hist)
{
    const bool firstSolve = !m->gamg && getenv("LDU_VERBOSE");
    const auto tSolve0 = std::chrono::steady_clock::now();
    if (ensure_hierarchy(m, c, true)) return -1;
    const auto tSolve1 = std::chrono::steady_clock::now();
    struct FirstSolveNote {
        bool on; std::chrono::steady_clock::time_point t0, t1; hipStream_t s;
        ~FirstSolveNote()
        {
            if (!on) return;
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "[ldugpu] first GAMG solve: hierarchy and level coefficients %.3f s, V-cycles (lazy engine plans "
                            "included) %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
        }
    } firstSolveNote{firstSolve, tSolve0, tSolve1, m->a->ctx->stream};
    GamgHierarchy* g = m->gamg;
    ldu_ctx* ctx = m->a->ctx;
    hipStream_t s = ctx->stream;
    const int n = m->a->nCells;
    double* Apsi = g->d_Apsi;
    double* finestCorrection = g->d_finestCorr;
    double* finestResidual = g->d_finestRes;

    if (dev_amul(m, Apsi, psi, false)) return -1;
    // normFactor uses finestCorrection as its temporary (:46-50)
    {
        if (dev_sumA(m, finestCorrection)) return -1;
        if (k_reduce(ctx, n, RED_SUM, psi, nullptr, nullptr, nullptr, S_SUMPSI, s)) return -1;
        const double cnt = (double)n;
        LDU_CHECK_HIP(hipMemcpyAsync(ctx->S() + S_COUNT, &cnt, sizeof(double), hipMemcpyHostToDevice, s));
        // (the level coefficients: launched from here, beside the finest level's first kernels)
        if (refresh_level_coeffs(m, c)) return -1;
        LDU_CHECK_HIP(hipStreamSynchronize(s));
        if (comm_allreduce_scalars(ctx, S_SUMPSI, 1, s)) return -1;
        if (comm_allreduce_scalars(ctx, S_COUNT, 1, s)) return -1;
        if (k_reduce(ctx, n, RED_NORMFACTOR, Apsi, source, finestCorrection, nullptr, S_NORM, s)) return -1;
        if (comm_allreduce_scalars(ctx, S_NORM, 1, s)) return -1;
    }
    if (k_ew(n, EW_SUB, finestResidual, source, Apsi, s)) return -1;
    if (k_reduce(ctx, n, RED_SUMMAG, finestResidual, nullptr, nullptr, nullptr, S_RES, s)) return -1;
    if (comm_allreduce_scalars(ctx, S_RES, 1, s)) return -1;
    double v[2];
    if (dev_read_scalars(ctx, S_RES, 2, v)) return -1;
    const double normFactor = v[1] + kSmall;
    perf->normFactor = normFactor;
    perf->initialResidual = v[0] / normFactor;
    perf->finalResidual = perf->initialResidual;
    if (hist && perf->nHistory < c->historyCapacity) hist[perf->nHistory] = perf->finalResidual;
    perf->nHistory++;

    if (!check_convergence(perf, c->tolerance, c->relTol))
    {
        do
        {
            if (vcycle(m, c, psi, source, Apsi, finestCorrection, finestResidual)) return -1;
            if (dev_amul(m, Apsi, psi, false)) return -1;
            // finestResidual = source; finestResidual -= Apsi (:96-98)
            if (k_ew(n, EW_SUB, finestResidual, source, Apsi, s)) return -1;
            if (k_reduce(ctx, n, RED_SUMMAG, finestResidual, nullptr, nullptr, nullptr, S_RES, s)) return -1;
            if (comm_allreduce_scalars(ctx, S_RES, 1, s)) return -1;
            double r;
            if (dev_read_scalars(ctx, S_RES, 1, &r)) return -1;
            perf->finalResidual = r / normFactor;
            if (hist && perf->nHistory < c->historyCapacity) hist[perf->nHistory] = perf->finalResidual;
            perf->nHistory++;
        } while (++perf->nIterations < c->maxIter && !check_convergence(perf, c->tolerance, c->relTol));
    }
    // (a solve that needed no V-cycle: whoever uses the matrices next on the main stream must find the side stream's fills done)
    if (g->aggPending && gamg_join_levels(g, ctx)) return -1;
    return 0;
}

// GAMGPreconditioner.C:44-128
int gamg_precondition_setup(ldu_matrix* m, const ldu_controls* c) { return ensure_hierarchy(m, c); }

int gamg_precondition(ldu_matrix* m, const ldu_controls* c, double* wA, const double* rA)
{
    GamgHierarchy* g = m->gamg;
    hipStream_t s = m->a->ctx->stream;
    const int n = m->a->nCells;
    double* AwA = g->d_Apsi;
    if (k_ew(n, EW_ZERO, wA, nullptr, nullptr, s)) return -1;
    if (k_ew(n, EW_COPY, g->d_finestRes, rA, nullptr, s)) return -1;
    for (int cycle = 0; cycle < c->nVcycles; cycle++)
    {
        if (vcycle(m, c, wA, rA, AwA, g->d_finestCorr, g->d_finestRes)) return -1;
        if (cycle < c->nVcycles - 1)
        {
            if (dev_amul(m, AwA, wA, false)) return -1;
            if (k_ew(n, EW_SUB, g->d_finestRes, rA, AwA, s)) return -1;
        }
    }
    return 0;
}

// ---------------------------------------------------------------- introspection (tests)

int gamg_build_for_query(ldu_matrix* m, const ldu_controls* c) { return ensure_hierarchy(m, c); }

int gamg_query(ldu_matrix* m, int32_t* nLevels, int32_t* nCells, int32_t* nFaces)
{
    GamgHierarchy* g = m->gamg;
    if (!g) { ldu_set_error("no GAMG hierarchy"); return -9; }
    *nLevels = (int)g->levels.size();
    for (size_t i = 0; i < g->levels.size(); i++)
    {
        nCells[i] = g->levels[i].addr->nCells;
        nFaces[i] = g->levels[i].addr->nFaces;
    }
    return 0;
}

// per-level facts for measurement reports: cells, faces, dependency levels, widest row, the engine that serves
// one GaussSeidel sweep / k pipelined sweeps on that level's addressing
int gamg_level_info(ldu_matrix* m, int level, int32_t out[8])
{
    GamgHierarchy* g = m->gamg;
    if (!g || level < 0 || level >= (int)g->levels.size()) { ldu_set_error("bad GAMG level"); return -9; }
    ldu_addr* a = g->levels[level].addr;
    out[0] = a->nCells; out[1] = a->nFaces; out[2] = a->nLevels; out[3] = a->maxRowWidth;
    out[4] = k_engine_of(a, 0); out[5] = k_engine_of(a, 1); out[6] = k_engine_of(a, 2); out[7] = a->nSlices;
    return 0;
}

int gamg_level_data(ldu_matrix* m, int level, int32_t* restrictAddr, double* diag, double* upper,
                    double* lower)
{
    GamgHierarchy* g = m->gamg;
    if (!g || level < 0 || level >= (int)g->levels.size()) { ldu_set_error("bad GAMG level"); return -9; }
    GamgLevel& L = g->levels[level];
    if (gamg_join_levels(g, m->a->ctx)) return -1;
    LDU_CHECK_HIP(hipStreamSynchronize(m->a->ctx->stream));
    if (restrictAddr) memcpy(restrictAddr, L.restrictAddr.data(), sizeof(int) * L.restrictAddr.size());
    if (diag) LDU_CHECK_HIP(hipMemcpy(diag, L.mat->d_diagO, sizeof(double) * L.addr->nCells, hipMemcpyDeviceToHost));
    if (upper && L.addr->nFaces)
        LDU_CHECK_HIP(hipMemcpy(upper, L.mat->d_upperO, sizeof(double) * L.addr->nFaces, hipMemcpyDeviceToHost));
    if (lower && L.addr->nFaces)
        LDU_CHECK_HIP(hipMemcpy(lower, L.mat->d_lowerO, sizeof(double) * L.addr->nFaces, hipMemcpyDeviceToHost));
    return 0;
}
