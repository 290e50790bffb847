// Cluster sweep engine ("row blocking"): fewer, fatter hand-offs.
//
// The level-scheduled engines pay one cross-CU hand-off (1.0-1.4 us) per dependency LEVEL: 646 of them on
// the 216^3 box.  Here the cells are grouped into compact clusters of <= 64 cells (one wavefront); the
// dependencies INSIDE a cluster are resolved by the wave itself, step by step through LDS (~0.15 us per
// step, no other wave involved), and only dependencies BETWEEN clusters go through the granule hand-off.
// For 4x4x4 blocks of a hex mesh: 10 internal steps per cluster, 163 cluster levels instead of 646.
//
// Clusters must form an acyclic quotient graph.  Any topological order of the cell DAG cut into
// consecutive chunks has that property (edges only run forward in the order), so the clusters are grown
// greedily IN a topological order: a cluster only ever absorbs "ready" cells (all lower neighbours already
// placed), preferring the ready cell with most neighbours inside the cluster (compact blobs; first come
// first served on ties, which yields cubes on structured numberings).
//
// The arithmetic per row is untouched (same accumulation order, -ffp-contract=off): bit-identical results.
// The engine is a secondary structure of an addressing: its own row order, entry table and granules;
// vectors stay in the level-ordered numbering and are reached through a row map.
#include <algorithm>
#include <map>
#include <queue>
#include <vector>

#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <thread>

#include "ldu_internal.hpp"
#include "ldu_cluster_greedy.hpp"

#ifndef CL_BLK
#define CL_BLK 256
#endif
#define CL_WPB (CL_BLK / LDU_WAVE)
#ifndef CL_NAP
#define CL_NAP 2      // x64 clocks between two polls of the dependencies of this sweep
#endif
#ifndef CL_NAP_UP
#define CL_NAP_UP 1   // ... of the previous sweep's values (pipelined GaussSeidel)
#endif
// Ticket counters.  One counter for the whole chip caps the engine at ~70 M tickets/s (one device-scope atomic on
// one address every ~14 ns: 216^3, 157 464 clusters = 39 366 tickets per sweep -> 0.55 ms, whatever else is done).
// CL_NQ counters on their own cache lines, workgroup b draws from counter b % CL_NQ (normally its XCD) and counter
// q hands out the chunks q, q + CL_NQ, ...: every counter still hands out its chunks in ascending order, and the
// lowest unfinished chunk is either held or the next ticket of its counter, so the sweep cannot deadlock as long
// as each counter has one workgroup that gets to run (grids this small to use one counter only).
#ifndef CL_NQ
#define CL_NQ 8
#endif
#define CL_QSTRIDE 32
struct ClBase { unsigned b[CL_NQ]; };
static inline int cl_nq(int grid) { return grid >= 8 * CL_NQ ? CL_NQ : 1; }
static inline void cl_advance(ClBase& B, int nChunks, int grid)
{
    const int nq = cl_nq(grid);
    for (int q = 0; q < nq; q++)
    {
        const int nCh = nChunks > q ? (nChunks - q + nq - 1) / nq : 0;
        const int nBl = grid > q ? (grid - q + nq - 1) / nq : 0;
        B.b[q] += (unsigned)(nCh + nBl);   // every workgroup overshoots its counter exactly once
    }
}
#define CL_MAXSEG 1536  // runs of the pipelined task list kept in LDS (12 KB)
#define CL_MAXD LDU_CL_MAXD     // dependencies (lower resp. upper neighbours) per row held in registers (variants 3 / 6 / 12)
#define CL_SPIN_LIMIT_DEFAULT (1u << 22)

typedef unsigned int cl_u32x4 __attribute__((ext_vector_type(4)));

// bound of every dependency wait, in polls (run-time overridable: ldu_ctx_set_spin_limit / LDU_SPIN_LIMIT)
__device__ unsigned g_cl_spin_limit = CL_SPIN_LIMIT_DEFAULT;
int k_cluster_set_spin_limit(unsigned polls)
{
    if (!polls) polls = CL_SPIN_LIMIT_DEFAULT;
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_cl_spin_limit), &polls, sizeof(unsigned)));
    return 0;
}

// Debug: per-task timeline of the pipelined GaussSeidel cluster sweeps (ldu_debug_cluster_trace): 8 x u64 per (sweep,
// cluster): tStart, tUpperDone, tLowerDone (poll success), tStepsDone, tStored [100 MHz wall clock], polls, XCC id, spare.
__device__ unsigned long long* g_cl_trace = nullptr;
int k_cluster_set_trace(unsigned long long* buf)
{
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_cl_trace), &buf, sizeof(buf)));
    return 0;
}

struct ClusterPlan {
    int nSlices = 0;
    long nEntries = 0;
    int nClusterLevels = 0;
    double avgDepth = 0;
    bool eligible = false;
    int maxDep = CL_MAXD;             // max over rows of max(nL, nU): picks the kernel instantiation
    // entries of cluster s start at s * fixedW * 64 when every cluster is (nearly) equally wide: a task then needs no
    // sliceEnt load before it can fetch its columns and coefficients (one dependent round trip less); 0 = variable
    int fixedW = 0;
    // cluster s owns the cluster-rows [64 s, 64 s + 64): fixed stride, so a wave finds its rows from the
    // ticket alone (one dependent load round trip less than with a row table); unused lanes carry intra = 255
    int* d_sliceEnt = nullptr;        // [nSlices]
    unsigned char* d_sliceDepth = nullptr;   // [nSlices] internal steps
    int2* d_rowMeta = nullptr;        // [64 nSlices] {level-ordered row, nL | nU << 8 | internal level << 16}
    long nRows = 0;                   // 64 nSlices
    int* d_colF = nullptr;            // [nEntries] lower part: cluster-row of the column; upper part: level row
    int* d_colB = nullptr;            // [nEntries] lower part: level row; upper part: cluster-row
    int* d_src = nullptr;             // [nEntries] index of the entry in the level-ordered SELL arrays
    int* d_srcFace = nullptr;         // [nEntries] face << 1 | (1: the upper-triangle coefficient of an owner row), -1: padding
    uint4* d_granule = nullptr;       // [nCells+1]
    unsigned* d_ticket = nullptr;
    ClBase ticketBase{};
    unsigned epoch = 0;
    int gen = 0;
    // second lane: a concurrent sweep of the same addressing on the second stream (PBiCG's transposed system)
    uint4* d_granule1 = nullptr;
    unsigned* d_ticket1 = nullptr;
    ClBase ticketBase1{};
    unsigned epoch1 = 0;
    int gen1 = 0;
    // component planes of the coupled (LduMatrix<Type,scalar,scalar>) sweeps: 3 granule planes, own tickets
    // (two lanes: the transposed system of PBiCCCG / PBiCICG runs concurrently on the second stream)
    uint4* d_granuleV[2] = {nullptr, nullptr};      // [3][nRows+1]
    unsigned* d_ticketV[2] = {nullptr, nullptr};
    ClBase ticketBaseV[2]{};
    unsigned epochV[2] = {0, 0};
    int genV[2] = {0, 0};
    std::vector<int> levelStart;      // [nClusterLevels+1] clusters of one cluster level are contiguous
    std::vector<int> upLevel;         // [nClusterLevels] running max of the cluster level holding an upper neighbour
    struct Tasks { int* d = nullptr; int n = 0; int* d_segStart = nullptr; int* d_segInfo = nullptr; int nSeg = 0; };
    std::map<int, Tasks> tasks;       // k -> topological (sweep, cluster) list of k pipelined GaussSeidel sweeps
    struct Conv { double* d = nullptr; unsigned long long stamp = 0; };
    std::map<const double*, Conv> conv;   // level-layout value array -> cluster-layout copy
};

template <class T>
static int cl_upload(T** dst, const std::vector<T>& src)
{
    const size_t n = src.size() ? src.size() : 1;
    LDU_CHECK_HIP(hipMalloc((void**)dst, n * sizeof(T)));
    if (src.size()) LDU_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int cl_upload_n(int** dst, const int* src, size_t n)
{
    LDU_CHECK_HIP(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(int)));
    if (n) LDU_CHECK_HIP(hipMemcpy(*dst, src, n * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

void cluster_free(ldu_addr* a)
{
    if (a->greedyThread.joinable()) a->greedyThread.join();
    delete a->greedyEarly;
    a->greedyEarly = nullptr;
    ClusterPlan* P = a->cluster;
    if (!P) return;
    void* ptrs[] = {P->d_sliceEnt, P->d_sliceDepth, P->d_rowMeta,
                    P->d_colF, P->d_colB, P->d_src, P->d_srcFace, P->d_granule, P->d_ticket, P->d_granule1, P->d_ticket1,
                    P->d_granuleV[0], P->d_ticketV[0], P->d_granuleV[1], P->d_ticketV[1]};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& kv : P->conv) if (kv.second.d) (void)hipFree(kv.second.d);
    for (auto& kv : P->tasks)
    {
        if (kv.second.d) (void)hipFree(kv.second.d);
        if (kv.second.d_segStart) (void)hipFree(kv.second.d_segStart);
        if (kv.second.d_segInfo) (void)hipFree(kv.second.d_segInfo);
    }
    delete P;
    a->cluster = nullptr;
}

void cluster_forget(ldu_addr* a, const double* levelVal)
{
    if (!a || !a->cluster) return;
    auto it = a->cluster->conv.find(levelVal);
    if (it == a->cluster->conv.end()) return;
    (void)hipStreamSynchronize(a->ctx->stream);
    if (it->second.d) (void)hipFree(it->second.d);
    a->cluster->conv.erase(it);
}

// Builds the plan on first use.  Returns 0 and sets P->eligible.
static int cluster_build(ldu_addr* a)
{
    if (a->cluster) return 0;
    const auto tBuild0 = std::chrono::steady_clock::now();
    ClusterPlan* P = new ClusterPlan();
    a->cluster = P;
    const int nC = a->nCells, nF = a->nFaces;
    const std::vector<int>& l = a->l;
    const std::vector<int>& u = a->u;
    // eligibility: register-resident dependencies
    int maxDep = 0;
    for (int c = 0; c < nC; c++)
    {
        maxDep = std::max(maxDep, a->losortStart[c + 1] - a->losortStart[c]);
        maxDep = std::max(maxDep, a->ownerStart[c + 1] - a->ownerStart[c]);
    }
    if (maxDep > CL_MAXD) return 0;
    P->maxDep = maxDep;
    if (nC < 64) return 0;

    // ---- greedy clustering in a topological order (ldu_cluster_greedy.hpp)
    if (a->greedyThread.joinable()) a->greedyThread.join();   // started by plan_build (large addressings)
    std::unique_ptr<ClGreedy> early(a->greedyEarly);
    a->greedyEarly = nullptr;
    ClGreedy GRlocal;
    if (!early)
        cluster_greedy(nC, nF, l.data(), u.data(), a->losort.data(), a->losortStart.data(), a->ownerStart.data(),
                       a->level.data(), LDU_WAVE, GRlocal);
    const ClGreedy& GR = early ? *early : GRlocal;
    const std::vector<int>& cluster = GR.cluster;
    const std::vector<int>& intra = GR.intra;
    const std::vector<int>& memberStart = GR.memberStart;
    const std::vector<int>& memberCells = GR.memberCells;
    const std::vector<int>& cLevel = GR.cLevel;
    const std::vector<int>& cDepth = GR.cDepth;
    const int nCl = (int)GR.nClusters();
    const auto tGreedy = std::chrono::steady_clock::now();
    // ---- schedule order: by cluster level (ties: creation order) = a topological order of the quotient
    std::vector<int> order(nCl);
    for (int i = 0; i < nCl; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cLevel[x] < cLevel[y]; });
    P->nSlices = nCl;
    int maxLev = 0; double sumDepth = 0;
    for (int i = 0; i < nCl; i++) { maxLev = std::max(maxLev, cLevel[i]); sumDepth += cDepth[i]; }
    P->nClusterLevels = maxLev + 1;
    P->avgDepth = sumDepth / std::max(1, nCl);

    std::vector<int> sliceEnt(nCl, 0), crowOf(nC);
    std::vector<unsigned char> sliceDepth(nCl);
    std::vector<int2> rowMeta((size_t)nCl * LDU_WAVE);
    long ent = 0;
    std::vector<int> sliceW(nCl);
    int Wmax = 0;
    // the clusters write disjoint ranges of every table: host threads over cluster ranges
    const int nT = nCl >= 4096 ? (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1;
    auto overClusters = [&](const std::function<void(int, int)>& fn) {
        if (nT == 1) { fn(0, nCl); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < nT; t++) th.emplace_back(fn, (int)((long)nCl * t / nT), (int)((long)nCl * (t + 1) / nT));
        for (auto& t : th) t.join();
    };
    overClusters([&](int s0, int s1) {
    for (int s = s0; s < s1; s++)
    {
        const int id = order[s];
        sliceDepth[s] = (unsigned char)cDepth[id];
        int W = 0;
        int row = s * LDU_WAVE;
        for (int q = memberStart[id]; q < memberStart[id + 1]; q++)
        {
            const int c = memberCells[q];
            crowOf[c] = row;
            const int cl_ = a->losortStart[c + 1] - a->losortStart[c], cu = a->ownerStart[c + 1] - a->ownerStart[c];
            rowMeta[row] = make_int2(a->iperm[c], cl_ | (cu << 8) | (intra[c] << 16) | (cDepth[id] << 24));
            W = std::max(W, cl_ + cu);
            row++;
        }
        // every lane of the cluster carries its depth (bits 24..31): the wave needs no sliceDepth load
        for (; row < (s + 1) * LDU_WAVE; row++) rowMeta[row] = make_int2(0, (255 << 16) | (cDepth[id] << 24));
        sliceW[s] = W;
    }
    });
    for (int s = 0; s < nCl; s++)
    {
        Wmax = std::max(Wmax, sliceW[s]);
        ent += (long)sliceW[s] * LDU_WAVE;
    }
    // fixed stride when the padding it costs is small (a hex mesh tiled by cubes: none)
    if (nCl && (double)Wmax * LDU_WAVE * nCl <= 1.15 * (double)ent && !getenv("LDU_CLUSTER_VARW")) P->fixedW = Wmax;
    ent = 0;
    for (int s = 0; s < nCl; s++)
    {
        if (P->fixedW) sliceW[s] = P->fixedW;
        sliceEnt[s] = (int)ent;
        ent += (long)sliceW[s] * LDU_WAVE;
        if (ent > 2000000000L) return 0;
    }
    P->nRows = (long)nCl * LDU_WAVE;
    P->nEntries = ent + 1024;
    {
        // cluster-level ranges and, per level, the highest cluster level that holds an upper neighbour
        P->levelStart.assign(P->nClusterLevels + 1, 0);
        for (int s = 0; s < nCl; s++) P->levelStart[cLevel[order[s]] + 1]++;
        for (int L = 0; L < P->nClusterLevels; L++) P->levelStart[L + 1] += P->levelStart[L];
        P->upLevel.assign(P->nClusterLevels, 0);
        {
            const int nTf = nT > 1 ? nT : 1;
            std::vector<std::vector<int>> part(nTf, std::vector<int>(P->nClusterLevels, 0));
            auto job = [&](int t) {
                std::vector<int>& up = part[t];
                for (long f = (long)nF * t / nTf; f < (long)nF * (t + 1) / nTf; f++)
                {
                    const int La = cLevel[cluster[l[f]]], Lb = cLevel[cluster[u[f]]];
                    if (Lb > up[La]) up[La] = Lb;
                }
            };
            if (nTf == 1) job(0);
            else
            {
                std::vector<std::thread> th;
                for (int t = 0; t < nTf; t++) th.emplace_back(job, t);
                for (auto& t : th) t.join();
            }
            for (int t = 0; t < nTf; t++)
                for (int L = 0; L < P->nClusterLevels; L++) P->upLevel[L] = std::max(P->upLevel[L], part[t][L]);
        }
        for (int L = 0; L < P->nClusterLevels; L++)
        {
            if (P->upLevel[L] < L) P->upLevel[L] = L;
            if (L && P->upLevel[L] < P->upLevel[L - 1]) P->upLevel[L] = P->upLevel[L - 1];
        }
    }
    // position of every (level-row, entry k) in the level-ordered SELL arrays
    std::vector<int> lvlSliceOfRow(nC), lvlSliceRow, lvlSliceEnt;
    {
        lvlSliceRow.resize(a->nSlices + 1); lvlSliceEnt.resize(a->nSlices);
        LDU_CHECK_HIP(hipMemcpy(lvlSliceRow.data(), a->d_sliceRow, sizeof(int) * (a->nSlices + 1), hipMemcpyDeviceToHost));
        LDU_CHECK_HIP(hipMemcpy(lvlSliceEnt.data(), a->d_sliceEnt, sizeof(int) * a->nSlices, hipMemcpyDeviceToHost));
        const int nS = a->nSlices;
        auto job = [&](int t, int n) {
            for (int s = (int)((long)nS * t / n); s < (int)((long)nS * (t + 1) / n); s++)
                for (int r = lvlSliceRow[s]; r < lvlSliceRow[s + 1]; r++) lvlSliceOfRow[r] = s;
        };
        if (nT == 1) job(0, 1);
        else
        {
            std::vector<std::thread> th;
            for (int t = 0; t < nT; t++) th.emplace_back(job, t, nT);
            for (auto& t : th) t.join();
        }
    }
    // (every entry is written below - the tables are not value-initialised first: 4 x 0.24 GB at 216^3)
    const size_t nEnt = (size_t)P->nEntries;
    std::unique_ptr<int[]> colF(new int[nEnt]), colB(new int[nEnt]), src(new int[nEnt]), srcFace(new int[nEnt]);
    for (size_t e = (size_t)ent; e < nEnt; e++) { colF[e] = 0; colB[e] = 0; src[e] = 0; srcFace[e] = -1; }
    auto fillRange = [&](int s0, int s1) {
    for (int s = s0; s < s1; s++)
    {
        const int id = order[s];
        for (int i = memberStart[id + 1] - memberStart[id]; i < LDU_WAVE; i++)   // lanes without a cell
            for (int k = 0; k < sliceW[s]; k++)
            {
                const long e = (long)sliceEnt[s] + i + (long)k * LDU_WAVE;
                colF[e] = 0; colB[e] = 0; src[e] = 0; srcFace[e] = -1;
            }
        for (int i = 0; i < memberStart[id + 1] - memberStart[id]; i++)
        {
            const int c = memberCells[memberStart[id] + i];
            const int r = s * LDU_WAVE + i;
            const int lr = a->iperm[c];
            const int ls = lvlSliceOfRow[lr];
            const long lbase = (long)lvlSliceEnt[ls] + (lr - lvlSliceRow[ls]);
            const long base = (long)sliceEnt[s] + (long)i;
            int k = 0;
            for (int j = a->losortStart[c]; j < a->losortStart[c + 1]; j++, k++)
            {
                const int nb = l[a->losort[j]];
                colF[base + (long)k * LDU_WAVE] = crowOf[nb];
                colB[base + (long)k * LDU_WAVE] = a->iperm[nb];
                src[base + (long)k * LDU_WAVE] = (int)(lbase + (long)k * LDU_WAVE);
                srcFace[base + (long)k * LDU_WAVE] = a->losort[j] << 1;
            }
            for (int f = a->ownerStart[c]; f < a->ownerStart[c + 1]; f++, k++)
            {
                const int nb = u[f];
                colF[base + (long)k * LDU_WAVE] = a->iperm[nb];
                colB[base + (long)k * LDU_WAVE] = crowOf[nb];
                src[base + (long)k * LDU_WAVE] = (int)(lbase + (long)k * LDU_WAVE);
                srcFace[base + (long)k * LDU_WAVE] = (f << 1) | 1;
            }
            for (; k < sliceW[s]; k++)
            {
                colF[base + (long)k * LDU_WAVE] = r; colB[base + (long)k * LDU_WAVE] = r;
                src[base + (long)k * LDU_WAVE] = (int)lbase;
                srcFace[base + (long)k * LDU_WAVE] = -1;
            }
            (void)r;
        }
    }
    };
    overClusters(fillRange);
    const auto tTables = std::chrono::steady_clock::now();
    if (cl_upload(&P->d_sliceEnt, sliceEnt) || cl_upload(&P->d_sliceDepth, sliceDepth) || cl_upload(&P->d_rowMeta, rowMeta)
        || cl_upload_n(&P->d_colF, colF.get(), nEnt) || cl_upload_n(&P->d_colB, colB.get(), nEnt)
        || cl_upload_n(&P->d_src, src.get(), nEnt) || (a->nFaces < (1 << 30) && cl_upload_n(&P->d_srcFace, srcFace.get(), nEnt)))
        return -1;
    LDU_CHECK_HIP(hipMalloc((void**)&P->d_granule, sizeof(uint4) * (size_t)(P->nRows + 1)));
    LDU_CHECK_HIP(ldu_memset_sync(P->d_granule, 0, sizeof(uint4) * (size_t)(P->nRows + 1)));
    LDU_CHECK_HIP(hipMalloc((void**)&P->d_ticket, sizeof(unsigned) * CL_NQ * CL_QSTRIDE));
    LDU_CHECK_HIP(ldu_memset_sync(P->d_ticket, 0, sizeof(unsigned) * CL_NQ * CL_QSTRIDE));
    LDU_CHECK_HIP(hipDeviceSynchronize());
    P->gen = a->ctx->p2pGen;
    P->eligible = true;
    if (getenv("LDU_VERBOSE"))
        fprintf(stderr, "[ldugpu] cluster plan: %d cells -> %d clusters (avg %.1f cells, avg %.1f internal steps), "
                        "%d cluster levels (dependency levels: %d), built in %.3f s (greedy %.3f, tables %.3f, upload %.3f)\n",
                nC, nCl, (double)nC / std::max(1, nCl), P->avgDepth, P->nClusterLevels, a->nLevels,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild0).count(),
                std::chrono::duration<double>(tGreedy - tBuild0).count(),
                std::chrono::duration<double>(tTables - tGreedy).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tTables).count());
    return 0;
}

// ---------------------------------------------------------------- kernels

__global__ void __launch_bounds__(CL_BLK)
cl_convert_kernel(long n, const int* __restrict__ src, const double* __restrict__ in, double* __restrict__ out)
{
    for (long i = (long)blockIdx.x * CL_BLK + threadIdx.x; i < n; i += (long)gridDim.x * CL_BLK) out[i] = in[src[i]];
}

// the cluster layout straight from the face-ordered coefficients (same values as fill_sell + cl_convert; the gather has
// the locality of the clusters instead of the level order's: a cluster's faces lie together)
__global__ void __launch_bounds__(CL_BLK)
cl_fill_kernel(long n, const int* __restrict__ srcFace, const double* __restrict__ lowerO, const double* __restrict__ upperO,
               double* __restrict__ out)
{
    for (long i = (long)blockIdx.x * CL_BLK + threadIdx.x; i < n; i += (long)gridDim.x * CL_BLK)
    {
        const int code = srcFace[i];
        double v = 0.0;
        if (code >= 0) v = (code & 1) ? upperO[code >> 1] : lowerO[code >> 1];
        out[i] = v;
    }
}

__device__ __forceinline__ void cl_store(uint4* G, int row, double v, unsigned tag)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    cl_u32x4 d;
    d.x = (unsigned)b; d.y = tag; d.z = (unsigned)(b >> 32); d.w = tag;
    uint4* p = G + row;
    // s_nop 1: a VMEM store of more than 64 bits reads its upper data registers late; the hazard recognizer
    // does not look inside inline asm, so the wait states before the next VALU write to those registers are
    // spelled out.  gfx940+ needs TWO (what the compiler inserts after its own dwordx4 stores on gfx950 is
    // `s_nop 1`); with `s_nop 0` the next pointer computation (v_lshl_add_u64 into the store's z,w
    // registers) could still win the race and publish pointer bits as the second tag - a consumer then
    // spins until its bound (round-1 fuzz case 8723: three plane stores in a row, under load)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}

// Three polls in flight, one wait.  Only the lanes that have that dependency outside the cluster issue the
// load (e0..e2 -> exec masks): in a 4x4x4 cluster 48 of the 192 (lane, dependency) pairs are external, and
// the polls of the whole chip are what fills the L2 request queues.  The other lanes' registers keep garbage
// that the callers never look at.
__device__ __forceinline__ void cl_load3(const uint4* p0, const uint4* p1, const uint4* p2, bool e0, bool e1,
                                         bool e2, cl_u32x4& g0, cl_u32x4& g1, cl_u32x4& g2)
{
    const unsigned long long m0 = __ballot(e0), m1 = __ballot(e1), m2 = __ballot(e2);
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %3, exec\n\t"
        "s_and_b64 exec, %3, %7\n\t"
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "s_and_b64 exec, %3, %8\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "s_and_b64 exec, %3, %9\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "s_mov_b64 exec, %3\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&s"(sv)
        : "v"(p0), "v"(p1), "v"(p2), "s"(m0), "s"(m1), "s"(m2)
        : "memory", "scc");
}

// nine granules in flight (three dependencies x three component planes), one wait
__device__ __forceinline__ void cl_load9(const uint4* p0, const uint4* p1, const uint4* p2, size_t gStride, bool e0,
                                         bool e1, bool e2, cl_u32x4 (&g)[3][3])
{
    const uint4 *q0 = p0 + gStride, *q1 = p1 + gStride, *q2 = p2 + gStride;
    const uint4 *r0 = q0 + gStride, *r1 = q1 + gStride, *r2 = q2 + gStride;
    const unsigned long long m0 = __ballot(e0), m1 = __ballot(e1), m2 = __ballot(e2);
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %9, exec\n\t"
        "s_and_b64 exec, %9, %19\n\t"
        "global_load_dwordx4 %0, %10, off sc1\n\t"
        "global_load_dwordx4 %3, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %16, off sc1\n\t"
        "s_and_b64 exec, %9, %20\n\t"
        "global_load_dwordx4 %1, %11, off sc1\n\t"
        "global_load_dwordx4 %4, %14, off sc1\n\t"
        "global_load_dwordx4 %7, %17, off sc1\n\t"
        "s_and_b64 exec, %9, %21\n\t"
        "global_load_dwordx4 %2, %12, off sc1\n\t"
        "global_load_dwordx4 %5, %15, off sc1\n\t"
        "global_load_dwordx4 %8, %18, off sc1\n\t"
        "s_mov_b64 exec, %9\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(g[0][0]), "=&v"(g[0][1]), "=&v"(g[0][2]), "=&v"(g[1][0]), "=&v"(g[1][1]), "=&v"(g[1][2]),
          "=&v"(g[2][0]), "=&v"(g[2][1]), "=&v"(g[2][2]), "=&s"(sv)
        : "v"(p0), "v"(p1), "v"(p2), "v"(q0), "v"(q1), "v"(q2), "v"(r0), "v"(r1), "v"(r2), "s"(m0), "s"(m1), "s"(m2)
        : "memory", "scc");
}

__device__ __forceinline__ double cl_value(const cl_u32x4& g)
{
    return __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
}

struct ClTab {
    const int* sliceEnt; const unsigned char* sliceDepth; const int2* rowMeta;
    const int* colDep;      // dependency part in cluster rows, the other part in level rows
    int fixedW;             // > 0: entries of cluster s start at s * fixedW * 64 (no sliceEnt load)
};
__device__ __forceinline__ long cl_ent0(const ClTab& T, int s)
{
    return T.fixedW ? (long)s * (long)(T.fixedW * LDU_WAVE) : (long)T.sliceEnt[s];
}

// MODE as SweepMode.  FWD modes: dependencies = lower part (entries 0..nl-1, ascending); BWD modes:
// dependencies = upper part, DEscending (TRI_BWD) resp. ascending (GS_BWD, symGaussSeidelSmoother.C:178-205).
template <int MODE, int ND>
__device__ __forceinline__ void cl_cluster(const ClTab& T, int s, int lane, double* __restrict__ lds,
                                           uint4* __restrict__ G, unsigned tag, volatile int* abortFlag,
                                           double* __restrict__ w, const double* __restrict__ rhs,
                                           const double* __restrict__ scale, const double* __restrict__ val,
                                           const double* __restrict__ val2, double* __restrict__ aux)
{
    constexpr int B = sw_base(MODE);
    constexpr bool TF = sw_tform(MODE);
    constexpr bool FWD = (B == SW_TRI_FWD || B == SW_RD || B == SW_GS_FWD);
    ldu_debug_stall(s == 0);
    const int row0 = s * LDU_WAVE;
    const int cnt = LDU_WAVE;
    const int r = row0 + lane;
    const int2 rm = T.rowMeta[r];
    const int depth = __builtin_amdgcn_readfirstlane(rm.y >> 24) & 255;
    const int myLv = (rm.y >> 16) & 255;
    const bool on = myLv != 255;
    const int lr = rm.x;
    const int nl = on ? (rm.y & 255) : 0, nu = on ? ((rm.y >> 8) & 255) : 0;
    const long ent = cl_ent0(T, s) + lane;
    const int nd = FWD ? nl : nu;            // dependencies
    const int d0 = FWD ? 0 : nl;             // first dependency entry
    // everything that does not depend on this sweep
    int c[ND];
    double v[ND], v2[ND];
#pragma unroll
    for (int k = 0; k < ND; k++)
    {
        const bool need = k < nd;
        const long e = ent + (long)(d0 + k) * LDU_WAVE;
        c[k] = need ? T.colDep[e] : r;
        v[k] = need ? val[e] : 0.0;
        v2[k] = (B == SW_RD && need) ? val2[e] : 0.0;
    }
    double acc, dd = 1.0;
    if (B == SW_TRI_FWD) { dd = scale[lr]; acc = dd * rhs[lr]; }
    else if (B == SW_TRI_BWD) { acc = w[lr]; if (TF) dd = scale[lr]; }
    else if (B == SW_RD) acc = scale[lr];
    else { acc = rhs[lr]; dd = scale[lr]; }
    // GaussSeidel divides by the diagonal: the denominator's half of the division is done before the wait
    const double rdd = ((B == SW_GS_FWD && !TF) || B == SW_GS_BWD) ? ldu_div_prepare(dd) : 0.0;
    double xu[ND], vu[ND];
    if (B == SW_GS_FWD)
    {
        // old values of the upper neighbours (level rows), loaded before anything of this sweep is written
#pragma unroll
        for (int k = 0; k < ND; k++)
        {
            const bool need = k < nu;
            const long e = ent + (long)(nl + k) * LDU_WAVE;
            vu[k] = need ? val[e] : 0.0;
            xu[k] = need ? w[T.colDep[e]] : 0.0;
        }
    }
    // external dependencies (other clusters): wait for all of them up front
    double xe[ND];
    bool internal[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) internal[k] = (k < nd) && (c[k] >= row0 && c[k] < row0 + cnt);
#pragma unroll
    for (int k0 = 0; k0 < ND; k0 += 3)
    {
        const bool e0 = (k0 < nd) && !internal[k0], e1 = (k0 + 1 < nd) && !internal[k0 + 1],
                   e2 = (k0 + 2 < nd) && !internal[k0 + 2];
        xe[k0] = 1.0; xe[k0 + 1] = 1.0; xe[k0 + 2] = 1.0;   // unused slots: finite, non-zero (coefficient 0)
        if (__any(e0 | e1 | e2))
        {
            cl_u32x4 g0, g1, g2;
            unsigned spins = 0;
            unsigned long long tw0 = 0;
            const unsigned spinLimit = g_cl_spin_limit;
            for (;;)
            {
                cl_load3(G + c[k0], G + c[k0 + 1], G + c[k0 + 2], e0, e1, e2, g0, g1, g2);
                bool ok = true;
                if (e0) ok &= (g0.y == tag) & (g0.w == tag);
                if (e1) ok &= (g1.y == tag) & (g1.w == tag);
                if (e2) ok &= (g2.y == tag) & (g2.w == tag);
                if (ok) break;
                if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0)) { *abortFlag = 1; return; }
                __builtin_amdgcn_s_sleep(CL_NAP);
            }
            if (e0) xe[k0] = cl_value(g0);
            if (e1) xe[k0 + 1] = cl_value(g1);
            if (e2) xe[k0 + 2] = cl_value(g2);
        }
    }
    // LDS image of what a row reads: slot i < 64 = value of cluster row i (written when it is computed),
    // slot 64 + k*64 + lane = this lane's k-th dependency when it comes from outside (or is unused).
    // Every lane then runs the same branch-free code in every step; only the rows of the step's internal
    // level commit.  Unused dependencies have coefficient 0 and read the finite 1.0: acc - 0*1 == acc.
    int slot[ND];
#pragma unroll
    for (int k = 0; k < ND; k++)
    {
        slot[k] = internal[k] ? c[k] - row0 : LDU_WAVE + k * LDU_WAVE + lane;
        lds[LDU_WAVE + k * LDU_WAVE + lane] = xe[k];
    }
    double pu[ND];
    if (B == SW_GS_FWD)
    {
#pragma unroll
        for (int k = 0; k < ND; k++) pu[k] = vu[k] * xu[k];   // 0*0 for unused entries
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double res = 0.0, auxv = 0.0;
    for (int st = 0; st < depth; st++)
    {
        const int lv = FWD ? st : depth - 1 - st;
        double t = acc;
        if (B == SW_TRI_BWD)
        {
            // DICPreconditioner.C:119-122: owned faces in DEscending order
#pragma unroll
            for (int k = ND - 1; k >= 0; k--)
            {
                if (TF) t -= dd * (v[k] * lds[slot[k]]);
                else t -= v[k] * lds[slot[k]];
            }
        }
        else if (B == SW_RD)
        {
#pragma unroll
            for (int k = 0; k < ND; k++)
            {
                if (TF) t -= (v2[k] * v[k]) * (1.0 / lds[slot[k]]);
                else t -= (v2[k] * v[k]) / lds[slot[k]];
            }
        }
        else if (B == SW_TRI_FWD && TF)
        {
#pragma unroll
            for (int k = 0; k < ND; k++) t -= dd * (v[k] * lds[slot[k]]);
        }
        else
        {
#pragma unroll
            for (int k = 0; k < ND; k++) t -= v[k] * lds[slot[k]];
        }
        if (myLv == lv)
        {
            double out = t;
            if (B == SW_GS_FWD)
            {
                auxv = t;
#pragma unroll
                for (int k = 0; k < ND; k++) t -= pu[k];   // unused: +0.0 (0 * 0), t - (+0.0) == t
                out = TF ? dd * t : ldu_div(t, dd, rdd);
            }
            else if (B == SW_GS_BWD) out = ldu_div(t, dd, rdd);
            lds[lane] = out;
            res = out;
        }
        // LDS is in order within a wave: make this step's values visible to the next step
        LDU_STEP_FENCE();
    }
    // publish the whole cluster at once: one coalesced granule store instead of a partial-line store per
    // step (PMC: 1.9x write amplification before).  Nothing is lost on the critical path: a consumer
    // cluster waits for ALL its external rows, and the last of them finishes in the last step anyway.
    if (on)
    {
        cl_store(G, r, res, tag);   // first what the neighbours wait for, then the scattered stores
        if (B == SW_GS_FWD && aux) aux[lr] = auxv;
        w[lr] = res;
    }
}

template <int MODE, bool DESC, int ND>
__global__ void __launch_bounds__(CL_BLK)
sweep_cluster_kernel(ClTab T, int nSlices, int nChunks, unsigned* ticket, ClBase ticketBase, uint4* G, unsigned tag,
                     int* abortFlag, double* w, const double* rhs, const double* scale, const double* val,
                     const double* val2, double* aux)
{
    __shared__ int s_chunk[2];
    __shared__ double s_x[CL_WPB][LDU_WAVE * (1 + ND)];   // slots 0..63: the cluster's rows, then ND x 64 outside values
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int nq = gridDim.x >= 8 * CL_NQ ? CL_NQ : 1;
    const int tq = blockIdx.x % nq;
    unsigned* const tk = ticket + tq * CL_QSTRIDE;
    const unsigned tb = ticketBase.b[tq];
    int nextT = 0;
    if (threadIdx.x == 0) nextT = (int)(atomicAdd(tk, 1u) - tb) * nq + tq;
    for (int it = 0;; it++)
    {
        if (threadIdx.x == 0)
        {
            const int t = ldu_abort_seen(abortFlag, it) ? 0x7fffffff : nextT;
            s_chunk[it & 1] = t;
            if (t < nChunks) nextT = (int)(atomicAdd(tk, 1u) - tb) * nq + tq;
        }
        __syncthreads();
        const int chunk = s_chunk[it & 1];
        if (chunk >= nChunks) return;
        const int si = chunk * CL_WPB + wave;
        if (si < nSlices)
            cl_cluster<MODE, ND>(T, DESC ? nSlices - 1 - si : si, lane, s_x[wave], G, tag, abortFlag, w, rhs, scale, val,
                             val2, aux);
    }
}

// level-layout value array -> cluster layout (cached until any coefficient array is rewritten)
static const double* cluster_values(ldu_addr* a, const double* levelVal, hipStream_t s)
{
    if (!levelVal) return nullptr;
    ClusterPlan* P = a->cluster;
    ClusterPlan::Conv& C = P->conv[levelVal];
    if (!C.d)
        if (hipMalloc((void**)&C.d, sizeof(double) * (size_t)P->nEntries) != hipSuccess) return nullptr;
    if (C.stamp != val_stamp(a, levelVal))
    {
        int grid = (int)std::min<long>((P->nEntries + CL_BLK - 1) / CL_BLK, 8192);
        auto org = a->valOrigin.find(levelVal);
        if (org != a->valOrigin.end() && P->d_srcFace && a->ctx->clusterDirectFill)
            cl_fill_kernel<<<grid, CL_BLK, 0, s>>>(P->nEntries - 1024, P->d_srcFace, org->second.first, org->second.second, C.d);
        else
            cl_convert_kernel<<<grid, CL_BLK, 0, s>>>(P->nEntries - 1024, P->d_src, levelVal, C.d);
        C.stamp = val_stamp(a, levelVal);
    }
    return C.d;
}

// (ahead of the first sweep, on the stream of the coefficient chain: see k_blocks_prefill)
int k_cluster_prefill(ldu_addr* a, const double* val, hipStream_t s)
{
    if (!a->cluster || !a->cluster->eligible) return 1;
    return cluster_values(a, val, s) ? 0 : -1;
}

template <int MODE, bool DESC>
static int launch_cluster(ldu_addr* a, const SweepArgs& g, hipStream_t s)
{
    ldu_ctx* ctx = a->ctx;
    ClusterPlan& P = *a->cluster;
    const double* val = cluster_values(a, g.val, s);
    const double* val2 = cluster_values(a, g.val2, s);
    if (!val) { ldu_set_error("cluster engine: value conversion failed"); return -1; }
    constexpr bool FWD = (sw_base(MODE) == SW_TRI_FWD || sw_base(MODE) == SW_RD || sw_base(MODE) == SW_GS_FWD);
    ClTab T{P.d_sliceEnt, P.d_sliceDepth, P.d_rowMeta, FWD ? P.d_colF : P.d_colB, P.fixedW};
    const int nChunks = (P.nSlices + CL_WPB - 1) / CL_WPB;
    // one workgroup per CU while a cluster level holds few clusters (fewer waiting waves: faster hand-offs),
    // two when it is wide (tools/det_probe.py: 64^3 .104 / .114 ms, 216^3 .763 / .603 ms at 1 / 2 per CU)
    int bpc = ctx->clusterBlocksPerCU;
    if (!ctx->clusterBpcForced && ctx->dualActive && bpc > 2) bpc = 2;   // two sweeps at once (PBiCG): 2 + 2 per CU
    if (!ctx->clusterBpcForced && P.nSlices < 150 * P.nClusterLevels) bpc = 1;
    int grid = ctx->numCUs * bpc;
    if (grid > nChunks) grid = nChunks;
    if (grid < 1) grid = 1;
    uint4* G = P.d_granule;
    unsigned* ticket = P.d_ticket;
    ClBase* base = &P.ticketBase;
    unsigned* epoch = &P.epoch;
    int* gen = &P.gen;
    if (g.lane == 1)
    {
        if (!P.d_granule1)
        {
            LDU_CHECK_HIP(hipMalloc((void**)&P.d_granule1, sizeof(uint4) * (size_t)(P.nRows + 1)));
            LDU_CHECK_HIP(ldu_memset_sync(P.d_granule1, 0, sizeof(uint4) * (size_t)(P.nRows + 1)));
            LDU_CHECK_HIP(hipMalloc((void**)&P.d_ticket1, sizeof(unsigned) * CL_NQ * CL_QSTRIDE));
            LDU_CHECK_HIP(ldu_memset_sync(P.d_ticket1, 0, sizeof(unsigned) * CL_NQ * CL_QSTRIDE));
            // hipMemset on device memory may return before the fill ran, and the compute streams do not wait
            // for the null stream: without this the first sweep can publish tags that the fill then erases
            LDU_CHECK_HIP(hipDeviceSynchronize());
            P.gen1 = ctx->p2pGen;
        }
        G = P.d_granule1; ticket = P.d_ticket1; base = &P.ticketBase1; epoch = &P.epoch1; gen = &P.gen1;
    }
    if (*gen != ctx->p2pGen)
    {
        LDU_CHECK_HIP(hipMemsetAsync(ticket, 0, sizeof(unsigned) * CL_NQ * CL_QSTRIDE, s));
        *base = ClBase{};
        *gen = ctx->p2pGen;
    }
    (*epoch)++;
    if (*epoch == 0) *epoch = 1;
    if (P.maxDep <= 3)
        sweep_cluster_kernel<MODE, DESC, 3><<<grid, CL_BLK, 0, s>>>(T, P.nSlices, nChunks, ticket, *base,
            G, *epoch, ctx->d_abort, g.w, g.rhs, g.scale, val, val2, g.aux);
    else if (P.maxDep <= 6)
        sweep_cluster_kernel<MODE, DESC, 6><<<grid, CL_BLK, 0, s>>>(T, P.nSlices, nChunks, ticket, *base,
            G, *epoch, ctx->d_abort, g.w, g.rhs, g.scale, val, val2, g.aux);
    else
        sweep_cluster_kernel<MODE, DESC, CL_MAXD><<<grid, CL_BLK, 0, s>>>(T, P.nSlices, nChunks, ticket, *base,
            G, *epoch, ctx->d_abort, g.w, g.rhs, g.scale, val, val2, g.aux);
    cl_advance(*base, nChunks, grid);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- component planes of the coupled family
// One cluster sweep for NC component planes of a Field<Type> (LduMatrix<Type, scalar, scalar>): column
// indices, coefficients, row metadata and the wait for the neighbouring clusters are shared by the planes,
// only the values are per plane.  B: SW_TRI_FWD / SW_TRI_BWD / SW_GS_FWD, always with the family's association
// (TDILUPreconditioner.C:108-124, TGaussSeidelSmoother.C:121-142): rD*(coeff*x), rD*acc.
template <int B, int ND, int NC>
__device__ __forceinline__ void cl_cluster_vec(const ClTab& T, int s, int lane, double* __restrict__ lds,
                                               uint4* __restrict__ G, size_t gStride, unsigned tag,
                                               volatile int* abortFlag, double* __restrict__ w,
                                               const double* __restrict__ rhs, size_t stride,
                                               const double* __restrict__ scale, const double* __restrict__ val)
{
    constexpr bool FWD = (B == SW_TRI_FWD || B == SW_GS_FWD);
    constexpr int LSZ = LDU_WAVE * (1 + ND);   // LDS doubles per plane
    const int row0 = s * LDU_WAVE;
    const int r = row0 + lane;
    const int2 rm = T.rowMeta[r];
    const int depth = __builtin_amdgcn_readfirstlane(rm.y >> 24) & 255;
    const int myLv = (rm.y >> 16) & 255;
    const bool on = myLv != 255;
    const int lr = rm.x;
    const int nl = on ? (rm.y & 255) : 0, nu = on ? ((rm.y >> 8) & 255) : 0;
    const long ent = cl_ent0(T, s) + lane;
    const int nd = FWD ? nl : nu;
    const int d0 = FWD ? 0 : nl;
    int c[ND];
    double v[ND];
#pragma unroll
    for (int k = 0; k < ND; k++)
    {
        const bool need = k < nd;
        const long e = ent + (long)(d0 + k) * LDU_WAVE;
        c[k] = need ? T.colDep[e] : r;
        v[k] = need ? val[e] : 0.0;
    }
    const double dd = scale[lr];
    double acc[NC];
#pragma unroll
    for (int j = 0; j < NC; j++)
    {
        if (B == SW_TRI_FWD) acc[j] = dd * rhs[j * stride + lr];
        else if (B == SW_TRI_BWD) acc[j] = w[j * stride + lr];
        else acc[j] = rhs[j * stride + lr];
    }
    double pu[NC][ND];
    if (B == SW_GS_FWD)
    {
        // old values of the upper neighbours (level rows), read before anything of this sweep is written
#pragma unroll
        for (int k = 0; k < ND; k++)
        {
            const bool need = k < nu;
            const long e = ent + (long)(nl + k) * LDU_WAVE;
            const double vu = need ? val[e] : 0.0;
            const int cu = need ? T.colDep[e] : lr;
#pragma unroll
            for (int j = 0; j < NC; j++) pu[j][k] = need ? vu * w[j * stride + cu] : 0.0;
        }
    }
    // slot < 64: the dependency is a row of this cluster; otherwise the lane's private copy of an outside value
    // (lane-varying flags are kept as these integers, not as booleans: a boolean per dependency and plane is
    // a 64-bit lane mask in scalar registers, and there are not enough of those at ND = 6)
    int slot[ND];
#pragma unroll
    for (int k = 0; k < ND; k++)
    {
        const bool in = (k < nd) && (c[k] >= row0 && c[k] < row0 + LDU_WAVE);
        slot[k] = in ? c[k] - row0 : LDU_WAVE + k * LDU_WAVE + lane;
    }
    // external dependencies: the three planes of a dependency are polled together (nine granules in flight
    // per round trip); the planes of a row are published together, so they normally arrive together
    static_assert(NC == 3, "three planes");
#pragma unroll
    for (int k0 = 0; k0 < ND; k0 += 3)
    {
        const bool e0 = (k0 < nd) && slot[k0] >= LDU_WAVE, e1 = (k0 + 1 < nd) && slot[k0 + 1] >= LDU_WAVE,
                   e2 = (k0 + 2 < nd) && slot[k0 + 2] >= LDU_WAVE;
        double x[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) { x[j][0] = 1.0; x[j][1] = 1.0; x[j][2] = 1.0; }   // unused: finite, coefficient 0
        if (__any(e0 | e1 | e2))
        {
            cl_u32x4 g[3][3];
            unsigned spins = 0;
            unsigned long long tw0 = 0;
            const unsigned spinLimit = g_cl_spin_limit;
            for (;;)
            {
                cl_load9(G + c[k0], G + c[k0 + 1], G + c[k0 + 2], gStride, e0, e1, e2, g);
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 3; j++)
                {
                    if (e0) ok &= (g[j][0].y == tag) & (g[j][0].w == tag);
                    if (e1) ok &= (g[j][1].y == tag) & (g[j][1].w == tag);
                    if (e2) ok &= (g[j][2].y == tag) & (g[j][2].w == tag);
                }
                if (ok) break;
                if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0)) { *abortFlag = 1; return; }
                __builtin_amdgcn_s_sleep(CL_NAP);
            }
#pragma unroll
            for (int j = 0; j < 3; j++)
            {
                if (e0) x[j][0] = cl_value(g[j][0]);
                if (e1) x[j][1] = cl_value(g[j][1]);
                if (e2) x[j][2] = cl_value(g[j][2]);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++)
        {
            lds[j * LSZ + LDU_WAVE + k0 * LDU_WAVE + lane] = x[j][0];
            lds[j * LSZ + LDU_WAVE + (k0 + 1) * LDU_WAVE + lane] = x[j][1];
            lds[j * LSZ + LDU_WAVE + (k0 + 2) * LDU_WAVE + lane] = x[j][2];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double res[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) res[j] = 0.0;
    for (int st = 0; st < depth; st++)
    {
        const int lv = FWD ? st : depth - 1 - st;
#pragma unroll
        for (int j = 0; j < NC; j++)
        {
            double t = acc[j];
            if (B == SW_TRI_BWD)
            {
#pragma unroll
                for (int k = ND - 1; k >= 0; k--) t -= dd * (v[k] * lds[j * LSZ + slot[k]]);
            }
            else if (B == SW_TRI_FWD)
            {
#pragma unroll
                for (int k = 0; k < ND; k++) t -= dd * (v[k] * lds[j * LSZ + slot[k]]);
            }
            else
            {
#pragma unroll
                for (int k = 0; k < ND; k++) t -= v[k] * lds[j * LSZ + slot[k]];
            }
            if (myLv == lv)
            {
                double out = t;
                if (B == SW_GS_FWD)
                {
#pragma unroll
                    for (int k = 0; k < ND; k++) t -= pu[j][k];   // unused entries hold 0.0: t - 0.0 == t
                    out = dd * t;
                }
                lds[j * LSZ + lane] = out;
                res[j] = out;
            }
        }
        LDU_STEP_FENCE();
    }
    if (on)
    {
#pragma unroll
        for (int j = 0; j < NC; j++)
        {
            cl_store(G + j * gStride, r, res[j], tag);
            w[j * stride + lr] = res[j];
        }
    }
}

template <int B, bool DESC, int ND, int NC>
__global__ void __launch_bounds__(CL_BLK)
sweep_cluster_vec_kernel(ClTab T, int nSlices, int nChunks, unsigned* ticket, ClBase ticketBase, uint4* G,
                         size_t gStride, unsigned tag, int* abortFlag, double* w, const double* rhs, size_t stride,
                         const double* scale, const double* val)
{
    __shared__ int s_chunk[2];
    __shared__ double s_x[CL_WPB][NC * LDU_WAVE * (1 + ND)];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    int nextT = 0;
    const int nq = gridDim.x >= 8 * CL_NQ ? CL_NQ : 1;
    const int tq = blockIdx.x % nq;
    unsigned* const tk = ticket + tq * CL_QSTRIDE;
    const unsigned tb = ticketBase.b[tq];
    if (threadIdx.x == 0) nextT = (int)(atomicAdd(tk, 1u) - tb) * nq + tq;
    for (int it = 0;; it++)
    {
        if (threadIdx.x == 0)
        {
            const int t = ldu_abort_seen(abortFlag, it) ? 0x7fffffff : nextT;
            s_chunk[it & 1] = t;
            if (t < nChunks) nextT = (int)(atomicAdd(tk, 1u) - tb) * nq + tq;
        }
        __syncthreads();
        const int chunk = s_chunk[it & 1];
        if (chunk >= nChunks) return;
        const int si = chunk * CL_WPB + wave;
        if (si < nSlices)
            cl_cluster_vec<B, ND, NC>(T, DESC ? nSlices - 1 - si : si, lane, s_x[wave], G, gStride, tag, abortFlag, w, rhs,
                                      stride, scale, val);
    }
}

static bool cluster_pays(const ldu_addr* a, int kind);

template <int B, bool DESC>
static int launch_cluster_vec(ldu_addr* a, double* w, const double* rhs, size_t stride, const double* scale,
                              const double* levelVal, int lane, hipStream_t s)
{
    ldu_ctx* ctx = a->ctx;
    ClusterPlan& P = *a->cluster;
    const double* val = cluster_values(a, levelVal, s);
    if (!val) { ldu_set_error("cluster engine: value conversion failed"); return -1; }
    constexpr bool FWD = (B == SW_TRI_FWD || B == SW_GS_FWD);
    ClTab T{P.d_sliceEnt, P.d_sliceDepth, P.d_rowMeta, FWD ? P.d_colF : P.d_colB, P.fixedW};
    const int nChunks = (P.nSlices + CL_WPB - 1) / CL_WPB;
    int bpc = ctx->clusterBlocksPerCU;
    if (!ctx->clusterBpcForced && ctx->dualActive && bpc > 2) bpc = 2;   // two sweeps at once (PBiCG): 2 + 2 per CU
    if (!ctx->clusterBpcForced && P.nSlices < 150 * P.nClusterLevels) bpc = 1;
    int grid = ctx->numCUs * bpc;
    if (grid > nChunks) grid = nChunks;
    if (grid < 1) grid = 1;
    const size_t gStride = (size_t)P.nRows + 1;
    if (!P.d_granuleV[lane])
    {
        LDU_CHECK_HIP(hipMalloc((void**)&P.d_granuleV[lane], sizeof(uint4) * 3 * gStride));
        LDU_CHECK_HIP(ldu_memset_sync(P.d_granuleV[lane], 0, sizeof(uint4) * 3 * gStride));
        LDU_CHECK_HIP(hipMalloc((void**)&P.d_ticketV[lane], sizeof(unsigned) * CL_NQ * CL_QSTRIDE));
        LDU_CHECK_HIP(ldu_memset_sync(P.d_ticketV[lane], 0, sizeof(unsigned) * CL_NQ * CL_QSTRIDE));
        LDU_CHECK_HIP(hipDeviceSynchronize());   // see d_granule1: the fill must have run before the first sweep
        P.genV[lane] = ctx->p2pGen;
    }
    if (P.genV[lane] != ctx->p2pGen)
    {
        LDU_CHECK_HIP(hipMemsetAsync(P.d_ticketV[lane], 0, sizeof(unsigned) * CL_NQ * CL_QSTRIDE, s));
        P.ticketBaseV[lane] = ClBase{};
        P.genV[lane] = ctx->p2pGen;
    }
    P.epochV[lane]++;
    if (P.epochV[lane] == 0) P.epochV[lane] = 1;
    if (P.maxDep <= 3)
        sweep_cluster_vec_kernel<B, DESC, 3, 3><<<grid, CL_BLK, 0, s>>>(T, P.nSlices, nChunks, P.d_ticketV[lane], P.ticketBaseV[lane],
            P.d_granuleV[lane], gStride, P.epochV[lane], ctx->d_abort, w, rhs, stride, scale, val);
    else
        sweep_cluster_vec_kernel<B, DESC, 6, 3><<<grid, CL_BLK, 0, s>>>(T, P.nSlices, nChunks, P.d_ticketV[lane], P.ticketBaseV[lane],
            P.d_granuleV[lane], gStride, P.epochV[lane], ctx->d_abort, w, rhs, stride, scale, val);
    cl_advance(P.ticketBaseV[lane], nChunks, grid);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// Three component planes (w, rhs: plane p at + p*stride) in one cluster sweep.  mode: SW_TRI_FWD_T,
// SW_TRI_BWD_T or SW_GS_FWD_T.  Returns 1 when the cluster engine does not take it (caller: plane by plane).
int k_sweep_cluster_vec3(ldu_addr* a, int mode, double* w, const double* rhs, size_t stride, const double* scale,
                         const double* val, int lane, hipStream_t s)
{
    ldu_ctx* ctx = a->ctx;
    if (!ctx->clusterEngine || !ctx->sweepP2P || a->nCells < ctx->clusterMinCells) return 1;
    if (cluster_build(a) < 0) return -1;
    if (!a->cluster->eligible || a->cluster->maxDep > 6) return 1;
    if (!cluster_pays(a, mode == SW_GS_FWD_T ? 1 : 0)) return 1;
    if (!s) s = ctx->stream;
    if (lane < 0 || lane > 1) return 1;
    switch (mode)
    {
    case SW_TRI_FWD_T: return launch_cluster_vec<SW_TRI_FWD, false>(a, w, rhs, stride, scale, val, lane, s);
    case SW_TRI_BWD_T: return launch_cluster_vec<SW_TRI_BWD, true>(a, w, rhs, stride, scale, val, lane, s);
    case SW_GS_FWD_T:  return launch_cluster_vec<SW_GS_FWD, false>(a, w, rhs, stride, scale, val, lane, s);
    }
    return 1;
}


// Build the cluster plans of several addressings at once (one host thread each): the greedy clustering is
// sequential per addressing but the levels of a GAMG hierarchy are independent.
// the cluster plan of ONE addressing, where the engine would consider it (callable from a set-up thread of its own)
int k_cluster_build_one(ldu_addr* a)
{
    if (!a || a->cluster || !a->ctx->clusterEngine || !a->ctx->sweepP2P || a->nCells < a->ctx->clusterMinCells) return 0;
    return cluster_build(a) < 0 ? -1 : 0;
}

int k_cluster_prebuild(const std::vector<ldu_addr*>& addrs)
{
    std::vector<ldu_addr*> todo;
    for (ldu_addr* a : addrs)
        if (a && !a->cluster && a->ctx->clusterEngine && a->ctx->sweepP2P && a->nCells >= a->ctx->clusterMinCells)
            todo.push_back(a);
    if (todo.size() < 2) return 0;
    std::vector<int> rc(todo.size(), 0);
    std::vector<std::string> err(todo.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < todo.size(); i++)
        th.emplace_back([&, i]() {
            if (hipSetDevice(todo[i]->ctx->device) != hipSuccess) { rc[i] = -1; err[i] = "hipSetDevice failed"; return; }
            rc[i] = cluster_build(todo[i]) < 0 ? -1 : 0;
            if (rc[i]) err[i] = ldu_last_error_string();
        });
    for (auto& t : th) t.join();
    for (size_t i = 0; i < todo.size(); i++)
        if (rc[i]) { ldu_set_error("cluster plan: " + err[i]); return -1; }
    return 0;
}

// returns 1 when the cluster engine does not take this sweep
int k_sweep_cluster(ldu_addr* a, const SweepArgs& g, hipStream_t s)
{
    ldu_ctx* ctx = a->ctx;
    if (!ctx->clusterEngine || a->nCells < ctx->clusterMinCells || g.lane > 1) return 1;
    if (cluster_build(a) < 0) return -1;
    if (!a->cluster->eligible) return 1;
    const int bm = sw_base(g.mode);
    if (!cluster_pays(a, (bm == SW_GS_FWD || bm == SW_GS_BWD) ? 1 : 0)) return 1;
    switch (g.mode)
    {
    case SW_TRI_FWD: return launch_cluster<SW_TRI_FWD, false>(a, g, s);
    case SW_TRI_BWD: return launch_cluster<SW_TRI_BWD, true>(a, g, s);
    case SW_RD:      return launch_cluster<SW_RD, false>(a, g, s);
    case SW_GS_FWD:  return launch_cluster<SW_GS_FWD, false>(a, g, s);
    case SW_GS_BWD:  return launch_cluster<SW_GS_BWD, true>(a, g, s);
    case SW_TRI_FWD_T: return launch_cluster<SW_TRI_FWD_T, false>(a, g, s);
    case SW_TRI_BWD_T: return launch_cluster<SW_TRI_BWD_T, true>(a, g, s);
    case SW_RD_T:      return launch_cluster<SW_RD_T, false>(a, g, s);
    case SW_GS_FWD_T:  return launch_cluster<SW_GS_FWD_T, false>(a, g, s);
    }
    return 1;
}

// Does the cluster engine pay for this addressing?  Hand-offs: clusterLevels x (1.4 us + internal steps x S)
// against levels x (1.4 us chip-wide | 1.0 us on narrow levels, where the slab engine runs).  S measured:
// ~0.1 us per step for the triangular sweeps, ~0.2 (3 dependencies) / 0.25 (6) for GaussSeidel (division).
// kind: 0 = triangular sweeps, 1 = one GaussSeidel sweep, 2 = pipelined GaussSeidel sweeps.
// Pipelined sweeps: with k sweeps in flight the level engines cost ~1.8-2.3 us per level, the cluster engine
// ~4-5 us per cluster level (GAMG hierarchy of the 216^3 box, round 2: wins at 2.6x fewer levels, loses at 1.8x; round 3,
// against the level engines as they are now: wins at 1.8x too - levels of 78 732 ... 19 683 cells, 0.250 / 0.190 / 0.149 ->
// 0.226 / 0.166 / 0.137 ms per 4 sweeps; factor 1.75, LDU_CLUSTER_PAYS).
static bool cluster_pays(const ldu_addr* a, int kind)
{
    const ClusterPlan& P = *a->cluster;
    if (a->ctx->clusterEngine > 1) return true;   // LDU_CLUSTER=2: forced
    if (kind == 2) return a->ctx->clusterPaysFactor * P.nClusterLevels <= a->nLevels;
    const double S = kind == 0 ? 0.1 : (P.maxDep <= 3 ? 0.2 : (P.maxDep <= 6 ? 0.25 : 0.35));
    const double perLevel = (a->nSlabs > 0 && a->slabWidth <= (kind == 0 ? 24.0 : 16.0)) ? 1.0 : 1.4;
    return P.nClusterLevels * (1.4 + P.avgDepth * S) < 0.85 * a->nLevels * perLevel;
}

bool k_cluster_active(ldu_addr* a)
{
    ldu_ctx* ctx = a->ctx;
    if (!ctx->clusterEngine || !ctx->sweepP2P || a->nCells < ctx->clusterMinCells) return false;
    if (cluster_build(a) < 0) return false;
    return a->cluster->eligible && cluster_pays(a, 2);
}

// ---------------------------------------------------------------- k pipelined GaussSeidel sweeps on clusters
// Same idea as sweep_p2p_gs_multi_kernel: sweep j+1 of a row needs sweep j+1's values of its lower
// neighbours and sweep j's values of its upper neighbours, so sweep j+1 trails sweep j by the cluster-level
// distance to the upper neighbours.  Tag tag0+j marks "value of sweep j"; only the last sweep writes psi.
// A row's sweep-j value is still in its granule when its lower neighbours read it in sweep j+1: the row
// itself cannot run sweep j+1 before those lower neighbours have published theirs.
template <int ND>
__device__ __forceinline__ void cl_gs_task(const ClTab& T, const int* __restrict__ colUp, int s, int j, int k,
                                           int traceStride, int lane, double* __restrict__ lds, uint4* __restrict__ G, unsigned tag0,
                                           volatile int* abortFlag, double* __restrict__ psi,
                                           const double* __restrict__ rhs, const double* __restrict__ diag,
                                           const double* __restrict__ val)
{
    ldu_debug_stall(s == 0 && j == 0);
    const int row0 = s * LDU_WAVE;
    const int cnt = LDU_WAVE;
    const int r = row0 + lane;
    const int2 rm = T.rowMeta[r];
    const int depth = __builtin_amdgcn_readfirstlane(rm.y >> 24) & 255;
    const int myLv = (rm.y >> 16) & 255;
    const bool on = myLv != 255;
    const int lr = rm.x;
    const int nl = on ? (rm.y & 255) : 0, nu = on ? ((rm.y >> 8) & 255) : 0;
    const long ent = cl_ent0(T, s) + lane;
    const unsigned tagNew = tag0 + (unsigned)j;
    unsigned long long* const trc = g_cl_trace ? g_cl_trace + ((size_t)j * (size_t)traceStride + (size_t)s) * 8 : nullptr;
    unsigned nPolls = 0;
    if (trc && lane == 0) trc[0] = (unsigned long long)wall_clock64();
    int c[ND], cu[ND];
    double v[ND], vu[ND];
#pragma unroll
    for (int q = 0; q < ND; q++)
    {
        const long e = ent + (long)q * LDU_WAVE;
        const long eu = ent + (long)(nl + q) * LDU_WAVE;
        c[q] = q < nl ? T.colDep[e] : r;
        v[q] = q < nl ? val[e] : 0.0;
        // upper neighbours: level rows (old psi, sweep 0) or cluster rows (granules of the previous sweep)
        cu[q] = q < nu ? (j == 0 ? T.colDep[eu] : colUp[eu]) : (j == 0 ? lr : r);
        vu[q] = q < nu ? val[eu] : 0.0;
    }
    const double acc0 = rhs[lr];
    const double dd = diag[lr];
    const double rdd = ldu_div_prepare(dd);   // the denominator's half of `t / dd`, before the waits
    double xu[ND];
    if (j == 0)
    {
#pragma unroll
        for (int q = 0; q < ND; q++) xu[q] = psi[cu[q]];
    }
    else
    {
        const unsigned t = tagNew - 1u;
#pragma unroll
        for (int k0 = 0; k0 < ND; k0 += 3)
        {
            xu[k0] = 0.0; xu[k0 + 1] = 0.0; xu[k0 + 2] = 0.0;
            const bool e0 = k0 < nu, e1 = k0 + 1 < nu, e2 = k0 + 2 < nu;
            if (__any(e0 | e1 | e2))
            {
                cl_u32x4 g0, g1, g2;
                unsigned spins = 0;
            unsigned long long tw0 = 0;
                const unsigned spinLimit = g_cl_spin_limit;
                for (;;)
                {
                    cl_load3(G + cu[k0], G + cu[k0 + 1], G + cu[k0 + 2], e0, e1, e2, g0, g1, g2);
                    bool ok = true;
                    if (e0) ok &= (g0.y == t) & (g0.w == t);
                    if (e1) ok &= (g1.y == t) & (g1.w == t);
                    if (e2) ok &= (g2.y == t) & (g2.w == t);
                    if (ok) break;
                    if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0)) { *abortFlag = 1; return; }
                    __builtin_amdgcn_s_sleep(CL_NAP_UP);
                }
                if (e0) xu[k0] = cl_value(g0);
                if (e1) xu[k0 + 1] = cl_value(g1);
                if (e2) xu[k0 + 2] = cl_value(g2);
            }
        }
    }
    if (trc && lane == 0) trc[1] = (unsigned long long)wall_clock64();
    double xe[ND];
    bool internal[ND];
#pragma unroll
    for (int q = 0; q < ND; q++) internal[q] = (q < nl) && (c[q] >= row0 && c[q] < row0 + cnt);
#pragma unroll
    for (int k0 = 0; k0 < ND; k0 += 3)
    {
        const bool e0 = (k0 < nl) && !internal[k0], e1 = (k0 + 1 < nl) && !internal[k0 + 1],
                   e2 = (k0 + 2 < nl) && !internal[k0 + 2];
        xe[k0] = 1.0; xe[k0 + 1] = 1.0; xe[k0 + 2] = 1.0;
        if (__any(e0 | e1 | e2))
        {
            cl_u32x4 g0, g1, g2;
            unsigned spins = 0;
            unsigned long long tw0 = 0;
            const unsigned spinLimit = g_cl_spin_limit;
            for (;;)
            {
                cl_load3(G + c[k0], G + c[k0 + 1], G + c[k0 + 2], e0, e1, e2, g0, g1, g2);
                nPolls++;
                bool ok = true;
                if (e0) ok &= (g0.y == tagNew) & (g0.w == tagNew);
                if (e1) ok &= (g1.y == tagNew) & (g1.w == tagNew);
                if (e2) ok &= (g2.y == tagNew) & (g2.w == tagNew);
                if (ok) break;
                if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0)) { *abortFlag = 1; return; }
                __builtin_amdgcn_s_sleep(CL_NAP);
            }
            if (e0) xe[k0] = cl_value(g0);
            if (e1) xe[k0 + 1] = cl_value(g1);
            if (e2) xe[k0 + 2] = cl_value(g2);
        }
    }
    int slot[ND];
    double pu[ND];
#pragma unroll
    for (int q = 0; q < ND; q++)
    {
        slot[q] = internal[q] ? c[q] - row0 : LDU_WAVE + q * LDU_WAVE + lane;
        lds[LDU_WAVE + q * LDU_WAVE + lane] = xe[q];
        pu[q] = q < nu ? vu[q] * xu[q] : 0.0;   // exactly +0.0 when unused (the step loop subtracts it unconditionally)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (trc && lane == 0) trc[2] = (unsigned long long)wall_clock64();
    double res = 0.0;
    for (int st = 0; st < depth; st++)
    {
        double t = acc0;
#pragma unroll
        for (int q = 0; q < ND; q++) t -= v[q] * lds[slot[q]];
        if (myLv == st)
        {
            // unused entries hold pu = 0 * 0 = +0.0, and t - (+0.0) == t for every t (also -0.0): no select
#pragma unroll
            for (int q = 0; q < ND; q++) t -= pu[q];
            const double out = ldu_div(t, dd, rdd);
            lds[lane] = out;
            res = out;
        }
        LDU_STEP_FENCE();
    }
    if (trc && lane == 0) trc[3] = (unsigned long long)wall_clock64();
    if (on)
    {
        // the granules are what the neighbours wait for: publish them before the (scattered) psi store
        cl_store(G, r, res, tagNew);
        if (j == k - 1) psi[lr] = res;
    }
    if (trc)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores are acknowledged
        if (lane == 0)
        {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            trc[4] = (unsigned long long)wall_clock64(); trc[5] = nPolls; trc[6] = xcc & 0xf; trc[7] = blockIdx.x;
        }
    }
}

template <int ND>
__global__ void __launch_bounds__(CL_BLK)
sweep_cluster_gs_multi_kernel(ClTab T, const int* __restrict__ colUp, const int* __restrict__ tasks, int nTasks,
                              const int* __restrict__ segStart, const int* __restrict__ segInfo, int nSeg,
                              int nChunks, int k, int nSlicesTrace, unsigned* ticket, ClBase ticketBase, uint4* G,
                              unsigned tag0, int* abortFlag, double* psi, const double* rhs, const double* diag,
                              const double* val)
{
    __shared__ int s_chunk[2];
    __shared__ double s_x[CL_WPB][LDU_WAVE * (1 + ND)];   // slots 0..63: the cluster's rows, then ND x 64 outside values
    // The task list is runs of consecutive clusters: (sweep j, cluster level L) = clusters levelStart[L] ... of sweep j.
    // With the run table in LDS a wave turns its ticket into a task with a few LDS reads instead of a global load of
    // tasks[ti] - one dependent memory round trip less between the ticket and the task's first poll.  (Tables with
    // more than CL_MAXSEG runs - very deep, irregular cluster DAGs - keep the global task list: nSeg = 0.)
    __shared__ int s_segStart[CL_MAXSEG + 1];
    __shared__ int s_segInfo[CL_MAXSEG];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < nSeg; i += CL_BLK) { s_segStart[i] = segStart[i]; s_segInfo[i] = segInfo[i]; }
    if (threadIdx.x == 0) s_segStart[nSeg] = nTasks;
    int cur = 0;
    int nextT = 0;
    const int nq = gridDim.x >= 8 * CL_NQ ? CL_NQ : 1;
    const int tq = blockIdx.x % nq;
    unsigned* const tk = ticket + tq * CL_QSTRIDE;
    const unsigned tb = ticketBase.b[tq];
    if (threadIdx.x == 0) nextT = (int)(atomicAdd(tk, 1u) - tb) * nq + tq;
    for (int it = 0;; it++)
    {
        if (threadIdx.x == 0)
        {
            const int t = ldu_abort_seen(abortFlag, it) ? 0x7fffffff : nextT;
            s_chunk[it & 1] = t;
            if (t < nChunks) nextT = (int)(atomicAdd(tk, 1u) - tb) * nq + tq;
        }
        __syncthreads();
        const int chunk = s_chunk[it & 1];
        if (chunk >= nChunks) return;
        const int ti = chunk * CL_WPB + wave;
        if (ti < nTasks)
        {
            int task;
            if (nSeg)
            {
                while (ti >= s_segStart[cur + 1]) cur++;      // tickets of a workgroup ascend: the cursor only moves forward
                const int info = s_segInfo[cur];
                task = info + (ti - s_segStart[cur]);         // (j << 28 | first cluster of the run) + offset in the run
            }
            else task = tasks[ti];
            cl_gs_task<ND>(T, colUp, task & 0x0fffffff, task >> 28, k, nSlicesTrace, lane, s_x[wave], G, tag0, abortFlag, psi, rhs, diag,
                       val);
        }
    }
}

// k pipelined GaussSeidel sweeps; returns 1 when the cluster engine does not take them
int k_sweep_cluster_gs_multi(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* valA)
{
    ldu_ctx* ctx = a->ctx;
    if (!k_cluster_active(a) || k < 1 || k > 7) return 1;
    ClusterPlan& P = *a->cluster;
    hipStream_t s = ctx->stream;
    auto it = P.tasks.find(k);
    if (it == P.tasks.end())
    {
        const int nLev = P.nClusterLevels;
        std::vector<int> tasks, segStart, segInfo;
        tasks.reserve((size_t)k * P.nSlices);
        std::vector<int> next(k, 0);
        bool progress = true;
        while (progress)
        {
            progress = false;
            // One round = the tasks that can run at the same time.  Sweep j may take cluster level L once sweep
            // j-1 has emitted every level <= upLevel[L] IN AN EARLIER ROUND (those tasks must have finished, not
            // just started): going through the sweeps from the last to the first makes next[j-1] the count
            // before this round.  With the sweeps ascending, a round held tasks that wait for each other, the
            // tasks runnable at one time lay (k-1) rounds apart and the workgroups' ticket window (grid chunks)
            // no longer covered them: k sweeps cost k/2 times two sweeps.
            for (int j = k - 1; j >= 0; j--)
            {
                const int L = next[j];
                if (L >= nLev) continue;
                if (j > 0 && next[j - 1] <= P.upLevel[L]) continue;
                if (P.levelStart[L + 1] > P.levelStart[L])
                {
                    segStart.push_back((int)tasks.size());
                    segInfo.push_back((j << 28) | P.levelStart[L]);
                }
                for (int sl = P.levelStart[L]; sl < P.levelStart[L + 1]; sl++) tasks.push_back((j << 28) | sl);
                next[j]++;
                progress = true;
            }
        }
        if (getenv("LDU_VERBOSE"))
        {
            int mx = 0; double sum = 0;
            for (int L = 0; L < nLev; L++) { mx = std::max(mx, P.upLevel[L] - L); sum += P.upLevel[L] - L; }
            fprintf(stderr, "[ldugpu] cluster multi-sweep order: %d cells, %d cluster levels, k = %d, sweep lag (upLevel - level) "
                            "avg %.1f max %d\n", a->nCells, nLev, k, sum / std::max(1, nLev), mx);
        }
        if (tasks.size() != (size_t)k * (size_t)P.nSlices)
        {
            ldu_set_error("cluster engine: the pipelined task order does not cover every (sweep, cluster) pair");
            return -1;
        }
        if (tasks.size() != (size_t)k * (size_t)P.nSlices)
        {
            ldu_set_error("cluster engine: the pipelined task order does not cover every (sweep, cluster) pair");
            return -1;
        }
        ClusterPlan::Tasks T;
        T.n = (int)tasks.size();
        LDU_CHECK_HIP(hipMalloc((void**)&T.d, sizeof(int) * (tasks.size() + 1)));
        LDU_CHECK_HIP(hipMemcpy(T.d, tasks.data(), sizeof(int) * tasks.size(), hipMemcpyHostToDevice));
        if (!segStart.empty() && (int)segStart.size() <= CL_MAXSEG && !getenv("LDU_CLUSTER_NOSEG"))
        {
            T.nSeg = (int)segStart.size();
            if (cl_upload(&T.d_segStart, segStart) || cl_upload(&T.d_segInfo, segInfo)) return -1;
        }
        it = P.tasks.emplace(k, T).first;
    }
    const double* val = cluster_values(a, valA, s);
    if (!val) { ldu_set_error("cluster engine: value conversion failed"); return -1; }
    ClTab T{P.d_sliceEnt, P.d_sliceDepth, P.d_rowMeta, P.d_colF, P.fixedW};
    const int nTasks = it->second.n;
    const int nChunks = (nTasks + CL_WPB - 1) / CL_WPB;
    int bpc = ctx->clusterBlocksPerCUMulti;
    int grid = ctx->numCUs * bpc;
    if (grid > nChunks) grid = nChunks;
    if (grid < 1) grid = 1;
    if (P.gen != ctx->p2pGen)
    {
        LDU_CHECK_HIP(hipMemsetAsync(P.d_ticket, 0, sizeof(unsigned) * CL_NQ * CL_QSTRIDE, s));
        P.ticketBase = ClBase{};
        P.gen = ctx->p2pGen;
    }
    if (P.epoch > 0xffffff00u)
    {
        LDU_CHECK_HIP(hipMemsetAsync(P.d_granule, 0, sizeof(uint4) * (size_t)(P.nRows + 1), s));
        P.epoch = 0;
    }
    const unsigned tag0 = P.epoch + 1;
    P.epoch += (unsigned)k;
    ctx->profStart(a, 4);
    if (P.maxDep <= 3)
        sweep_cluster_gs_multi_kernel<3><<<grid, CL_BLK, 0, s>>>(T, P.d_colB, it->second.d, nTasks,
            it->second.d_segStart, it->second.d_segInfo, it->second.nSeg, nChunks, k, P.nSlices,
            P.d_ticket, P.ticketBase, P.d_granule, tag0, ctx->d_abort, psi, rhs, diag, val);
    else if (P.maxDep <= 6)
        sweep_cluster_gs_multi_kernel<6><<<grid, CL_BLK, 0, s>>>(T, P.d_colB, it->second.d, nTasks,
            it->second.d_segStart, it->second.d_segInfo, it->second.nSeg, nChunks, k, P.nSlices,
            P.d_ticket, P.ticketBase, P.d_granule, tag0, ctx->d_abort, psi, rhs, diag, val);
    else
        sweep_cluster_gs_multi_kernel<CL_MAXD><<<grid, CL_BLK, 0, s>>>(T, P.d_colB, it->second.d, nTasks,
            it->second.d_segStart, it->second.d_segInfo, it->second.nSeg, nChunks, k, P.nSlices,
            P.d_ticket, P.ticketBase, P.d_granule, tag0, ctx->d_abort, psi, rhs, diag, val);
    ctx->profStop(a, 4);
    cl_advance(P.ticketBase, nChunks, grid);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// Debug: cluster levels of the plan (out[0] = clusters, out[1] = cluster levels, then levelStart[0..levels])
int k_cluster_levels(ldu_addr* a, int* out, int cap)
{
    if (!a->cluster || !a->cluster->eligible) { ldu_set_error("no cluster plan for this addressing"); return -1; }
    const ClusterPlan& P = *a->cluster;
    if (cap < 3 + P.nClusterLevels) { ldu_set_error("k_cluster_levels: buffer too small"); return -1; }
    out[0] = P.nSlices; out[1] = P.nClusterLevels;
    for (int L = 0; L <= P.nClusterLevels; L++) out[2 + L] = P.levelStart[L];
    return 0;
}

bool k_cluster_kind_active(ldu_addr* a, int kind)
{
    ldu_ctx* ctx = a->ctx;
    if (!ctx->clusterEngine || !ctx->sweepP2P || a->nCells < ctx->clusterMinCells) return false;
    if (cluster_build(a) < 0) return false;
    return a->cluster->eligible && cluster_pays(a, kind == 0 ? 0 : 1);
}

int k_cluster_set_watchdog(unsigned long long budgetTicks, unsigned long long stallTicks)
{
    const unsigned long long v[2] = {budgetTicks, stallTicks};
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wait_budget), v, sizeof(v)));
    return 0;
}
