// Device "plan" of one lduAddressing: dependency levels, level-ordered renumbering,
// sliced-ELL entry layout and the sweep schedule.  Built once per addressing
// (the reference builds losort/ownerStart lazily once per mesh: lduAddressing.C:31-169).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include "ldu_internal.hpp"
#include "ldu_cluster_greedy.hpp"

// Width class of a row (lower + upper neighbours): the rows of a dependency level are stored class by class.
//   0..3  narrow: at most 8 on each side (one lane per row: eight granule polls in flight cover the dependencies);
//         total <= 6 | <= 8 | <= 12 | <= 16
//   4..6  wide: more than 8 on one side, total <= 16 | <= 32 | <= 64 -> COOPERATIVE rows: 2 | 4 | 8 lanes per row
//         (each lane loads, polls and multiplies eight entries, one lane subtracts the products in face order)
//   7     more than 64 entries: one lane per row, serial (never seen on a mesh; the layout allows 255 + 255)
static inline int row_width_class(int cl, int cu)
{
    const int t = cl + cu;
    if (cl <= 8 && cu <= 8) return t <= 6 ? 0 : (t <= 8 ? 1 : (t <= 12 ? 2 : 3));
    return t <= 16 ? 4 : (t <= 32 ? 5 : (t <= 64 ? 6 : 7));
}
static inline int class_lanes(int cls) { return cls == 4 ? 2 : (cls == 5 ? 4 : (cls == 6 ? 8 : 1)); }

template <class T>
static int upload(T** dst, const std::vector<T>& src)
{
    size_t n = src.size() ? src.size() : 1;
    LDU_CHECK_HIP(hipMalloc((void**)dst, n * sizeof(T)));
    if (src.size())
        LDU_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

ldu_addr::P2PLane* ldu_addr::lane(int i)
{
    P2PLane& P = p2p[i];
    if (!P.d_granule)
    {
        // tags start at 0 = never published
        if (hipMalloc((void**)&P.d_granule, sizeof(uint4) * (size_t)(nCells + 1)) != hipSuccess) return nullptr;
        if (ldu_memset_sync(P.d_granule, 0, sizeof(uint4) * (size_t)(nCells + 1)) != hipSuccess) return nullptr;
        // [0] chunk tickets, [32] chunks reported complete (run-ahead window), each on its own cache line
        if (hipMalloc((void**)&P.d_ticket, sizeof(unsigned) * 64) != hipSuccess) return nullptr;
        if (ldu_memset_sync(P.d_ticket, 0, sizeof(unsigned) * 64) != hipSuccess) return nullptr;
        if (nSlabs > 0)
        {
            if (hipMalloc((void**)&P.d_X, sizeof(uint4) * (size_t)(nCells + 1)) != hipSuccess) return nullptr;
            if (ldu_memset_sync(P.d_X, 0, sizeof(uint4) * (size_t)(nCells + 1)) != hipSuccess) return nullptr;
            // [2][8] per-slab tickets, [2][8] per-slab completed chunks, double-buffered by launch parity
            if (hipMalloc((void**)&P.d_ctl, sizeof(unsigned) * 32) != hipSuccess) return nullptr;
            if (ldu_memset_sync(P.d_ctl, 0, sizeof(unsigned) * 32) != hipSuccess) return nullptr;
            P.par = 0;
        }
        // hipMemset on device memory may return before the fill ran and the compute streams do not wait for
        // the null stream: the fills must be complete before the first sweep publishes its tags
        if (hipDeviceSynchronize() != hipSuccess) return nullptr;
        P.ticketBase = 0;
        P.doneBase = 0;
        P.epoch = 0;
        P.gen = ctx->p2pGen;
    }
    return &P;
}

double* ldu_addr::scratchVec(int i)
{
    while ((int)scratch.size() <= i) scratch.push_back(nullptr);
    if (!scratch[i])
    {
        size_t n = (size_t)std::max(nCells, std::max(nFaces, 1)) * 3 + 64;
        if (hipMalloc((void**)&scratch[i], n * sizeof(double)) != hipSuccess) return nullptr;
    }
    return scratch[i];
}

// XCD slabs.  A hand-off between two workgroups of the SAME XCD can stay in that XCD's L2 (plain
// granule store, L1-bypassing load): measured 1.0 us per dependency level against 1.4 us for the
// chip-wide write-through hand-off (profiles/r01_xcd_slab_probe.md).  The cells are therefore cut
// into up to 8 contiguous ranges of the ORIGINAL numbering, one per XCD.  lower < upper for every
// face, so dependencies between slabs only run from a lower to a higher slab (forward sweeps; the
// reverse for backward sweeps): the slabs form a one-directional pipeline in which the slower
// cross-XCD hand-off is a start-up delay, not a per-level cost.
// Returns the slab count (0 = keep the chip-wide engine) and the slab boundaries (cell index).
static int choose_slabs(const ldu_addr* a, std::vector<int>& slabCell)
{
    const ldu_ctx* ctx = a->ctx;
    const int nC = a->nCells;
    slabCell.clear();
    if (!ctx->sweepP2P || ctx->p2pSlabs == 0 || ctx->nXcd <= 0 || ctx->p2pGate || nC == 0)
        return 0;
    const int wavesPerXcd = std::max(1, ctx->numCUs / ctx->nXcd) * ctx->p2pBlocksPerCU * 4;
    int S;
    if (ctx->p2pSlabs > 0)
        S = std::min(ctx->p2pSlabs, ctx->nXcd);
    else
    {
        // measured (tools/det_probe.py, profiles/r01_xcd_slab_probe.md): one XCD up to ~150k cells,
        // all of them above; intermediate counts never won
        // (irregular graphs whose rows are grouped by lag - many small slices per level - keep all XCDs busy down to
        //  ~12 k cells: octree twin's GAMG levels 6-9, 144 k ... 17 k cells, 4 sweeps: 2.7 / 1.9 / 1.6 / 1.4 ms on one
        //  slab, 2.1 / 1.4 / 1.3 / 1.1 ms on eight)
        S = nC <= (a->lagBuckets ? 12000 : 150000) ? 1 : ctx->nXcd;
    }
    // equal shares of the entries
    slabCell.assign(S + 1, nC);
    slabCell[0] = 0;
    {
        const long total = (long)nC + 2L * a->nFaces;
        long acc = 0;
        int s = 1;
        for (int c = 0; c < nC && s < S; c++)
        {
            acc += 1 + (a->losortStart[c + 1] - a->losortStart[c]) + (a->ownerStart[c + 1] - a->ownerStart[c]);
            if (acc * S >= total * s) slabCell[s++] = c + 1;
        }
    }
    const_cast<ldu_addr*>(a)->slabWidth = (double)nC / LDU_WAVE / std::max(1, a->nLevels) / S;
    if (ctx->p2pSlabs > 0) return S;
    // cost model, in hand-offs: a level costs one round per wavesPerXcd slices of its busiest slab
    // (slab engine) against 1.4 x one round per chip-load of slices (chip-wide engine).  Rejects
    // numberings whose index ranges follow the levels (then the slabs run one after the other).
    std::vector<int> w((size_t)a->nLevels * S, 0);
    for (int s = 0; s < S; s++)
        for (int c = slabCell[s]; c < slabCell[s + 1]; c++) w[(size_t)a->level[c] * S + s]++;
    double roundsSlab = 0, roundsChip = 0;
    for (int L = 0; L < a->nLevels; L++)
    {
        int mx = 0, tot = 0;
        for (int s = 0; s < S; s++) { mx = std::max(mx, w[(size_t)L * S + s]); tot += w[(size_t)L * S + s]; }
        const int slS = (mx + LDU_WAVE - 1) / LDU_WAVE, slC = (tot + LDU_WAVE - 1) / LDU_WAVE;
        roundsSlab += std::max(1, (slS + wavesPerXcd - 1) / wavesPerXcd);
        roundsChip += 1.4 * std::max(1, (slC + wavesPerXcd * ctx->nXcd - 1) / (wavesPerXcd * ctx->nXcd));
    }
    if (roundsSlab > roundsChip) { slabCell.clear(); return 0; }
    return S;
}

// host threads over [0, n) in contiguous ranges (the big table fills write disjoint ranges)
static void par_ranges(long n, long grain, const std::function<void(long, long)>& fn)
{
    const int nT = n >= grain ? (int)std::min<long>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), n / grain) : 1;
    if (nT <= 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nT; t++) th.emplace_back(fn, n * t / nT, n * (t + 1) / nT);
    for (auto& t : th) t.join();
}

int plan_build(ldu_addr* a)
{
    const int nC = a->nCells, nF = a->nFaces;
    // LDU_VERBOSE: where the plan of a large addressing spends its time
    struct Phases {
        bool on; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); std::string txt;
        void mark(const char* name)
        {
            if (!on) return;
            char b[64];
            snprintf(b, sizeof(b), " %s %.3f", name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            txt += b;
        }
    } ph{getenv("LDU_VERBOSE") != nullptr && nC >= 1000000};
    const std::vector<int>& l = a->l;
    const std::vector<int>& u = a->u;

    // validate: upper-triangular order by owner (lduAddressing.H:36-63)
    for (int f = 0; f < nF; f++)
    {
        if (l[f] < 0 || u[f] >= nC || l[f] >= u[f])
        {
            ldu_set_error("ldu_addr_create: face " + std::to_string(f) + " violates lower < upper < nCells");
            return -2;
        }
        if (f && l[f] < l[f - 1])
        {
            ldu_set_error("ldu_addr_create: faces must be sorted by owner (upper-triangular order)");
            return -2;
        }
    }

    // losort / ownerStart / losortStart (lduAddressing.C:31-169)
    a->losort.assign(nF, 0);
    a->ownerStart.assign(nC + 1, 0);
    a->losortStart.assign(nC + 1, 0);
    {
        std::vector<int> cnt(nC + 1, 0);
        for (int f = 0; f < nF; f++) cnt[u[f] + 1]++;
        for (int c = 0; c < nC; c++) cnt[c + 1] += cnt[c];
        for (int c = 0; c <= nC; c++) a->losortStart[c] = cnt[c];
        for (int f = 0; f < nF; f++) a->losort[cnt[u[f]]++] = f;
        std::vector<int> oc(nC + 1, 0);
        for (int f = 0; f < nF; f++) oc[l[f] + 1]++;
        for (int c = 0; c < nC; c++) oc[c + 1] += oc[c];
        a->ownerStart = oc;
    }

    ph.mark("losort");
    // dependency levels of the lower-triangular DAG: one pass in face order is enough
    // because every face into cell k (owner < k) precedes the faces owned by k.
    a->level.assign(nC, 0);
    for (int f = 0; f < nF; f++)
        a->level[u[f]] = std::max(a->level[u[f]], a->level[l[f]] + 1);
    int nLevels = 0;
    for (int c = 0; c < nC; c++) nLevels = std::max(nLevels, a->level[c] + 1);
    a->nLevels = nLevels;
    {
        // the cluster engine's greedy clustering needs the graph and the levels only: on large addressings it starts now,
        // on a thread of its own, beside the rest of this plan (216^3: 1.0 s next to the 1.0 s that follow here)
        const ldu_ctx* ctx = a->ctx;
        int maxDep = 0;
        for (int c = 0; c < nC; c++)
            maxDep = std::max(maxDep, std::max(a->losortStart[c + 1] - a->losortStart[c], a->ownerStart[c + 1] - a->ownerStart[c]));
        static const bool early = !getenv("LDU_NO_EARLY_GREEDY");
        if (early && ctx->clusterEngine && ctx->sweepP2P && nC >= std::max(ctx->clusterMinCells, 200000) && maxDep <= LDU_CL_MAXD
            && !a->greedyThread.joinable() && !a->greedyEarly && !a->cluster)
        {
            a->greedyEarly = new ClGreedy();
            a->greedyThread = std::thread([a, nC, nF]() {
                cluster_greedy(nC, nF, a->l.data(), a->u.data(), a->losort.data(), a->losortStart.data(), a->ownerStart.data(),
                               a->level.data(), LDU_WAVE, *a->greedyEarly);
            });
        }
    }
    ph.mark("levels");
    a->levelStart.assign(nLevels + 1, 0);
    for (int c = 0; c < nC; c++) a->levelStart[a->level[c] + 1]++;
    for (int L = 0; L < nLevels; L++) a->levelStart[L + 1] += a->levelStart[L];
    a->perm.assign(nC, 0);
    a->iperm.assign(nC, 0);
    {
        // Inside a level the rows are independent, so their order is free.  Narrow rows first in their original
        // order, then the wider ones class by class (stable): a slice is padded to its widest row, and on meshes
        // with hanging faces / agglomerated levels ~1 % wide rows scattered over the level would sit in half of its
        // slices (octree twin of the motorBike mesh: 50 % of the slices padded from 6 to 9..25 entries per row).
        // Classes: row_width_class above.
        const bool sortRows = a->ctx->sortRowsByWidth;
        auto widthClass = [&](int c) -> int {
            if (!sortRows) return 0;
            return row_width_class(a->losortStart[c + 1] - a->losortStart[c], a->ownerStart[c + 1] - a->ownerStart[c]);
        };
        // Lag buckets.  Pipelined GaussSeidel sweeps: sweep j+1 of a row needs sweep j's values of its UPPER neighbours,
        // so it trails sweep j by t1 - level, t1 = the row's dependency level in the two-sweep DAG.  On a mesh numbering
        // with locality that lag is the same everywhere (hex box: 2); on irregular graphs (bandCompression numberings
        // of octree meshes, agglomerated GAMG levels) most rows trail by a few levels and a few by hundreds - and one
        // such row holds back its whole slice (a wave waits for ALL its rows), and with it everything downstream of the
        // slice in the later sweep: k sweeps cost k times one (octree twin, GAMG level 3: 1124 steps for 4 sweeps
        // against 335 for one; 424 with every row on its own).  Rows of a level are therefore also grouped by
        // (t1 - level) / lagWidth: slices hold rows that become runnable together (same level 3: 696 steps).
        // (not for matrices the single-wavefront kernels take - up to smallMaxCells cells, rows up to 16 entries: they
        //  walk the slices one after the other, more slices only cost them)
        int widestRow = 0;
        for (int c = 0; c < nC; c++)
            widestRow = std::max(widestRow, a->losortStart[c + 1] - a->losortStart[c] + a->ownerStart[c + 1] - a->ownerStart[c]);
        // (nor for the levels of the one-workgroup engine: every slice is a task there, full slices are what it wants.
        //  LDU_WG_WIDE=1 lets it take rows of any width - measured on the octree twin's coarse levels it loses to the slab
        //  engine with cooperative rows and lag buckets there, 1.2-2.3 against 0.6-1.1 ms for four sweeps)
        //  Below ~400 cells it wins with rows of any width - there a level is a chain of one-slice steps and the hand-off is
        //  everything: levels of 77 / 152 / 303 cells of the real motorBike mesh 0.138 / 0.179 / 0.298 -> 0.089 / 0.111 / 0.231 ms
        //  for four sweeps, the 1200-cell level 0.41 -> 0.49 ms)
        a->wgLevel = a->ctx->wgEngine && nC <= std::min(a->ctx->wgMaxCells, 18000) && nC >= a->ctx->wgMinCells
                     && (widestRow <= 16 || a->ctx->wgWide || nC <= 400);
        const bool smallKernels = (nC <= a->ctx->smallMaxCells && widestRow <= 16) || a->wgLevel;
        const int lagW = (sortRows && a->ctx->lagBucketWidth > 0 && !smallKernels && nC >= 512) ? a->ctx->lagBucketWidth : 0;
        int NLAG = 1;
        std::vector<unsigned char> lagB(nC, 0);
        if (lagW)
        {
            std::vector<int> t1(nC, 0);
            for (int f = 0; f < nF; f++) t1[l[f]] = std::max(t1[l[f]], a->level[u[f]]);          // sweep 0's upper neighbours
            for (int c = 0; c < nC; c++) t1[c] = std::max(t1[c], a->level[c]) + 1;
            for (int f = 0; f < nF; f++) t1[u[f]] = std::max(t1[u[f]], t1[l[f]] + 1);             // sweep 1's lower neighbours
            int maxLag = 0;
            for (int c = 0; c < nC; c++) maxLag = std::max(maxLag, t1[c] - a->level[c]);
            if (maxLag > 2 * lagW)   // (a uniform lag - structured numberings - needs no buckets)
            {
                // linear buckets all the way (half-octave buckets above 32 levels were tried on the octree twin's GAMG
                // levels: 1072 instead of 696 steps for 4 sweeps of level 3 - the far-reaching rows matter too)
                NLAG = std::min(256, maxLag / lagW + 1);
                // (the counting sort below holds levels x classes x buckets counters: very deep DAGs get fewer buckets)
                while (NLAG > 1 && (size_t)nLevels * 8 * (size_t)NLAG > ((size_t)1 << 26)) NLAG /= 2;
                for (int c = 0; c < nC; c++) lagB[c] = (unsigned char)std::min((t1[c] - a->level[c]) / lagW, NLAG - 1);
                a->lagBuckets = true;
            }
        }
        constexpr int NCLS = 8;
        const int NKEY = NCLS * (a->lagBuckets ? NLAG : 1);
        std::vector<int> cnt((size_t)nLevels * NKEY + 1, 0);
        std::vector<unsigned short> cls(nC);
        a->rowKey.assign(nC, 0);
        for (int c = 0; c < nC; c++)
        {
            cls[c] = (unsigned short)(widthClass(c) * (a->lagBuckets ? NLAG : 1) + lagB[c]);
            cnt[(size_t)a->level[c] * NKEY + cls[c] + 1]++;
        }
        for (size_t i = 0; i + 1 < cnt.size(); i++) cnt[i + 1] += cnt[i];
        for (int c = 0; c < nC; c++)   // stable: original order inside a (level, class, lag bucket)
        {
            const int r = (int)cnt[(size_t)a->level[c] * NKEY + cls[c]]++;
            a->perm[r] = c;
            a->iperm[c] = r;
            a->rowKey[r] = cls[c];
        }
    }

    // XCD slabs (see choose_slabs): contiguous ranges of the ORIGINAL cell index.  Inside a level the rows are ordered by
    // (width class, lag bucket) first and by original index only within such a group, so rowSlab is non-decreasing inside a
    // (level, class, bucket) group, not inside the level: slices are cut wherever it changes (below), and on multi-slab
    // levels every group fragments at the slab boundaries
    ph.mark("perm");
    std::vector<int> slabCell;
    const int S = choose_slabs(a, slabCell);
    a->nSlabs = S;
    std::vector<unsigned char> rowSlab(nC, 0);
    if (S > 1)
        for (int s = 0; s < S; s++)
            for (int c = slabCell[s]; c < slabCell[s + 1]; c++) rowSlab[a->iperm[c]] = (unsigned char)s;

    // slices (<= 64 rows, never straddling a level or a slab)
    std::vector<int> sliceRow, sliceCnt, sliceEnt, sliceW, sliceSlab;
    a->levelSliceStart.assign(nLevels + 1, 0);
    std::vector<unsigned char> nL(nC), nU(nC);
    for (int r = 0; r < nC; r++)
    {
        int c = a->perm[r];
        int cl = a->losortStart[c + 1] - a->losortStart[c];
        int cu = a->ownerStart[c + 1] - a->ownerStart[c];
        if (cl > 255 || cu > 255)
        {
            ldu_set_error("ldu_addr_create: more than 255 lower or upper neighbours in one row");
            return -2;
        }
        nL[r] = (unsigned char)cl;
        nU[r] = (unsigned char)cu;
        a->maxUpper = std::max(a->maxUpper, cu);
    }
    std::vector<unsigned char> rowClass(nC, 0);
    for (int r = 0; r < nC; r++) rowClass[r] = (unsigned char)row_width_class(nL[r], nU[r]);
    // cooperative rows need the rows of a level grouped by class
    const bool coop = a->ctx->sortRowsByWidth && a->ctx->coopRows && !a->wgLevel;
    std::vector<unsigned char> sliceT;
    long ent = 0;
    for (int L = 0; L < nLevels; L++)
    {
        a->levelSliceStart[L] = (int)sliceRow.size();
        for (int r0 = a->levelStart[L], cnt = 0; r0 < a->levelStart[L + 1]; r0 += cnt)
        {
            cnt = 1;
            // (a slice also ends where the width class changes, on levels wide enough not to care about one more
            //  slice: the single-wavefront kernels of the small levels walk the slices one after the other)
            const bool cutAtClass = a->ctx->sortRowsByWidth && a->levelStart[L + 1] - a->levelStart[L] >= 256;
            // a cooperative slice holds rows of ONE wide class, 64 / lanes-per-row of them
            const int Tl = coop ? class_lanes(rowClass[r0]) : 1;
            const int maxCnt = LDU_WAVE / Tl;
            while (cnt < maxCnt && r0 + cnt < a->levelStart[L + 1] && rowSlab[r0 + cnt] == rowSlab[r0]
                   && !((cutAtClass || (coop && (Tl > 1 || class_lanes(rowClass[r0 + cnt]) > 1)))
                        && rowClass[r0 + cnt] != rowClass[r0])
                   && !(a->lagBuckets && a->rowKey[r0 + cnt] != a->rowKey[r0])) cnt++;
            sliceT.push_back((unsigned char)Tl);
            if (Tl > 1) a->nCoopSlices++;
            sliceSlab.push_back(rowSlab[r0]);
            int W = 0;
            for (int i = 0; i < cnt; i++) W = std::max(W, (int)nL[r0 + i] + (int)nU[r0 + i]);
            sliceRow.push_back(r0);
            sliceCnt.push_back(cnt);
            sliceEnt.push_back((int)ent);
            sliceW.push_back(W);
            a->maxRowWidth = std::max(a->maxRowWidth, W);
            ent += (long)W * LDU_WAVE;
            if (ent > 2000000000L)
            {
                ldu_set_error("ldu_addr_create: entry count exceeds int32 addressing");
                return -2;
            }
        }
    }
    a->levelSliceStart[nLevels] = (int)sliceRow.size();
    a->nSlices = (int)sliceRow.size();
    // 512 padding entries: the fast GaussSeidel path reads 8 entries per row unconditionally
    const long entPad = ent + 1024;   // (the small-matrix kernel reads 16)
    a->nEntries = entPad;
    sliceRow.push_back(nC);

    ph.mark("slices");
    std::vector<int> col((size_t)entPad, 0), face((size_t)entPad, -1);
    par_ranges(a->nSlices, 4096, [&](long s0, long s1) {
    for (long s = s0; s < s1; s++)
    {
        for (int i = 0; i < sliceCnt[s]; i++)
        {
            int r = sliceRow[s] + i;
            int c = a->perm[r];
            long base = (long)sliceEnt[s] + i;
            int k = 0;
            for (int j = a->losortStart[c]; j < a->losortStart[c + 1]; j++, k++)
            {
                int f = a->losort[j];
                col[base + (long)k * LDU_WAVE] = a->iperm[l[f]];
                face[base + (long)k * LDU_WAVE] = f;
            }
            for (int f = a->ownerStart[c]; f < a->ownerStart[c + 1]; f++, k++)
            {
                col[base + (long)k * LDU_WAVE] = a->iperm[u[f]];
                face[base + (long)k * LDU_WAVE] = f;
            }
            // padding columns point at the row itself (never dereferenced: k < nL+nU guards)
            for (; k < sliceW[s]; k++) col[base + (long)k * LDU_WAVE] = r;
        }
        for (int i = sliceCnt[s]; i < LDU_WAVE; i++)
            for (int k = 0; k < sliceW[s]; k++)
                col[(long)sliceEnt[s] + i + (long)k * LDU_WAVE] = sliceRow[s];
    }
    });
    ph.mark("col/face");

    // slab engine tables
    std::vector<int> slabList, colX;
    std::vector<unsigned char> xflag;
    if (S > 0)
    {
        slabList.reserve(a->nSlices);
        for (int sl = 0; sl < S; sl++)
        {
            a->slabStart[sl] = (int)slabList.size();
            for (int s = 0; s < a->nSlices; s++)
                if (sliceSlab[s] == sl) slabList.push_back(s);
        }
        for (int sl = S; sl <= 8; sl++) a->slabStart[sl] = (int)slabList.size();
        // dependency levels each slab spans: all of them on a natural numbering (slabs = slices across the level
        // planes), one eighth when the numbering follows the levels (bandCompression) - what the run-ahead window
        // of a slab's queue is measured in
        {
            std::vector<int> lo(S, nLevels), hi(S, -1);
            for (int L = 0; L < nLevels; L++)
                for (int s = a->levelSliceStart[L]; s < a->levelSliceStart[L + 1]; s++)
                {
                    lo[sliceSlab[s]] = std::min(lo[sliceSlab[s]], L);
                    hi[sliceSlab[s]] = std::max(hi[sliceSlab[s]], L);
                }
            for (int sl = 0; sl < 8; sl++) a->slabLevelSpan[sl] = sl < S && hi[sl] >= lo[sl] ? hi[sl] - lo[sl] + 1 : 1;
        }
        colX = col;
        xflag.assign(nC, 0);
        if (S > 1)
            par_ranges(a->nSlices, 4096, [&](long s0, long s1) {
            for (long s = s0; s < s1; s++)
                for (int i = 0; i < sliceCnt[s]; i++)
                {
                    const int r = sliceRow[s] + i;
                    for (int k = 0; k < (int)nL[r] + (int)nU[r]; k++)
                    {
                        const long e = (long)sliceEnt[s] + i + (long)k * LDU_WAVE;
                        const int q = col[e];
                        if (rowSlab[q] != rowSlab[r])
                        {
                            colX[e] = (int)((unsigned)q | 0x80000000u);
                            // (several threads may flag the same row: relaxed atomic stores of the same value)
                            __atomic_store_n(&xflag[r], (unsigned char)1, __ATOMIC_RELAXED);
                            __atomic_store_n(&xflag[q], (unsigned char)1, __ATOMIC_RELAXED);
                        }
                    }
                }
            });
    }
    ph.mark("slabs");

    // polling gates of the point-to-point sweeps: the slice one dependency level before the
    // latest-scheduled slice this one depends on (forward: lower neighbours; backward: upper)
    // (LDU_P2P_GATE experiment only: nothing reads them otherwise)
    std::vector<int> gateF(a->nSlices, -1), gateB(a->nSlices, -1);
    if (a->ctx->p2pGate)
    {
        std::vector<int> rowSlice(nC), sliceLevel(a->nSlices);
        for (int L = 0; L < nLevels; L++)
            for (int s = a->levelSliceStart[L]; s < a->levelSliceStart[L + 1]; s++)
            {
                sliceLevel[s] = L;
                for (int i = 0; i < sliceCnt[s]; i++) rowSlice[sliceRow[s] + i] = s;
            }
        for (int s = 0; s < a->nSlices; s++)
        {
            int depF = -1, depB = a->nSlices;
            for (int i = 0; i < sliceCnt[s]; i++)
            {
                const int r = sliceRow[s] + i;
                const long base = (long)sliceEnt[s] + i;
                for (int k = 0; k < nL[r]; k++)
                    depF = std::max(depF, rowSlice[col[base + (long)k * LDU_WAVE]]);
                for (int k = nL[r]; k < nL[r] + nU[r]; k++)
                    depB = std::min(depB, rowSlice[col[base + (long)k * LDU_WAVE]]);
            }
            if (depF >= 0)
            {
                const int L = sliceLevel[depF];
                const int g = depF - (a->levelSliceStart[L + 1] - a->levelSliceStart[L]);
                gateF[s] = g >= 0 ? g : -1;
            }
            if (depB < a->nSlices)
            {
                const int L = sliceLevel[depB];
                const int g = depB + (a->levelSliceStart[L + 1] - a->levelSliceStart[L]);
                gateB[s] = g < a->nSlices ? g : -1;
            }
        }
    }

    // sweep schedule: runs of small levels are fused into single-block chains
    a->segs.clear();
    {
        const int fuseRows = a->ctx->fuseRows;
        int L = 0;
        while (L < nLevels)
        {
            int rows = a->levelStart[L + 1] - a->levelStart[L];
            Segment sg;
            sg.levelBegin = L;
            if (rows <= fuseRows)
            {
                int E = L;
                while (E < nLevels && (a->levelStart[E + 1] - a->levelStart[E]) <= fuseRows) E++;
                sg.levelEnd = E;
                sg.fused = true;
            }
            else
            {
                sg.levelEnd = L + 1;
                sg.fused = false;
            }
            sg.sliceBegin = a->levelSliceStart[sg.levelBegin];
            sg.sliceEnd = a->levelSliceStart[sg.levelEnd];
            a->segs.push_back(sg);
            L = sg.levelEnd;
        }
    }

    ph.mark("gates/segments");
    if (upload(&a->d_perm, a->perm)) return -1;
    if (upload(&a->d_iperm, a->iperm)) return -1;
    if (upload(&a->d_sliceRow, sliceRow)) return -1;
    if (upload(&a->d_sliceCnt, sliceCnt)) return -1;
    if (upload(&a->d_sliceEnt, sliceEnt)) return -1;
    if (upload(&a->d_sliceW, sliceW)) return -1;
    if (upload(&a->d_sliceT, sliceT)) return -1;
    if (upload(&a->d_levelSliceStart, a->levelSliceStart)) return -1;
    if (upload(&a->d_nL, nL)) return -1;
    if (upload(&a->d_nU, nU)) return -1;
    if (upload(&a->d_col, col)) return -1;
    if (upload(&a->d_face, face)) return -1;
    if (upload(&a->d_l, a->l)) return -1;
    if (upload(&a->d_u, a->u)) return -1;
    if (upload(&a->d_losort, a->losort)) return -1;
    if (upload(&a->d_ownerStart, a->ownerStart)) return -1;
    if (upload(&a->d_losortStart, a->losortStart)) return -1;
    if (S > 0)
    {
        if (upload(&a->d_slabList, slabList)) return -1;
        if (upload(&a->d_colX, colX)) return -1;
        if (upload(&a->d_xflag, xflag)) return -1;
    }
    // point-to-point sweep state (tags start at 0 = never published)
    if (!a->lane(0)) return -1;
    if (upload(&a->d_gateF, gateF)) return -1;
    if (upload(&a->d_gateB, gateB)) return -1;
    LDU_CHECK_HIP(hipMalloc((void**)&a->d_sliceDone, sizeof(unsigned) * (size_t)(a->nSlices + 1)));
    LDU_CHECK_HIP(ldu_memset_sync(a->d_sliceDone, 0, sizeof(unsigned) * (size_t)(a->nSlices + 1)));
    ph.mark("upload");
    if (ph.on) fprintf(stderr, "[ldugpu] level plan of %d cells, cumulative seconds:%s\n", nC, ph.txt.c_str());
    return 0;
}

// Coupled patches: boundary-row lists in the reference's update order
// (lduMatrixUpdateMatrixInterfaces.C:96-266: patch by patch, face by face).
int plan_finalize_patches(ldu_addr* a)
{
    int off = 0;
    for (auto& p : a->patches) { p.offset = off; off += p.n; }
    a->nPatchFaces = off;
    if (off == 0) { a->finalized = true; return 0; }

    std::vector<int> pfCell(off);
    std::vector<std::vector<int>> perRow;   // by new row index -> list of patch-face ids
    std::map<int, int> rowSlot;
    std::vector<int> bRow;
    for (auto& p : a->patches)
    {
        for (int i = 0; i < p.n; i++)
        {
            int r = a->iperm[p.faceCells[i]];
            pfCell[p.offset + i] = r;
            auto it = rowSlot.find(r);
            if (it == rowSlot.end())
            {
                rowSlot[r] = (int)bRow.size();
                bRow.push_back(r);
                perRow.emplace_back();
                it = rowSlot.find(r);
            }
            perRow[it->second].push_back(p.offset + i);   // ascending (patch, face) by construction
        }
    }
    std::vector<int> bStart(bRow.size() + 1, 0), bFace;
    for (size_t i = 0; i < bRow.size(); i++)
    {
        bStart[i + 1] = bStart[i] + (int)perRow[i].size();
        bFace.insert(bFace.end(), perRow[i].begin(), perRow[i].end());
    }
    a->nBRows = (int)bRow.size();
    {
        // nonBlockingGaussSeidelSmoother.C:66-79: blockStart_ = smallest cell touched by a coupled patch
        int blockStart = a->nCells;
        for (auto& p : a->patches)
            for (int c : p.faceCells) blockStart = std::min(blockStart, c);
        std::vector<int> rowB(a->nCells, -1);
        std::vector<unsigned char> k0(a->nCells, 0);
        for (size_t i = 0; i < bRow.size(); i++) rowB[bRow[i]] = (int)i;
        for (int r = 0; r < a->nCells; r++)
        {
            const int c = a->perm[r];
            int k = 0;   // lower-part entries are in ascending face order = ascending owner
            for (int j = a->losortStart[c]; j < a->losortStart[c + 1]; j++)
                if (a->l[a->losort[j]] < blockStart) k++;
            k0[r] = (unsigned char)k;
        }
        if (a->d_nbRowB) { (void)hipFree(a->d_nbRowB); a->d_nbRowB = nullptr; }
        if (a->d_nbK0) { (void)hipFree(a->d_nbK0); a->d_nbK0 = nullptr; }
        if (upload(&a->d_nbRowB, rowB)) return -1;
        if (upload(&a->d_nbK0, k0)) return -1;
    }
    if (upload(&a->d_bRow, bRow)) return -1;
    if (upload(&a->d_bStart, bStart)) return -1;
    if (upload(&a->d_bFace, bFace)) return -1;
    if (upload(&a->d_pfCell, pfCell)) return -1;
    LDU_CHECK_HIP(hipMalloc((void**)&a->d_sendAll, sizeof(double) * (size_t)off));
    LDU_CHECK_HIP(hipMalloc((void**)&a->d_recvAll, sizeof(double) * (size_t)off));
    LDU_CHECK_HIP(ldu_memset_sync(a->d_recvAll, 0, sizeof(double) * (size_t)off));
    for (auto& p : a->patches)
    {
        p.d_send = a->d_sendAll + p.offset;
        p.d_recv = a->d_recvAll + p.offset;
        p.d_faceCells = a->d_pfCell + p.offset;
    }
    a->finalized = true;
    // peer-store backend: receive region in this rank's window, offsets exchanged with the neighbours (collective over
    // the ranks that share patches: every rank finalizes its addressings in the same order)
    return comm_peer_setup_addr(a);
}

thread_local bool tl_bgPlanThread = false;

void plan_free(ldu_addr* a)
{
    addr_bg_wait(a);      // (a sweep plan still being built behind the solves)
    comm_peer_free_addr(a);
    if (a->d_cycPair) { (void)hipFree(a->d_cycPair); a->d_cycPair = nullptr; }
    cluster_free(a);
    blocks_free(a);
    gs_layouts_free(a);
    for (auto& kv : a->graphs) (void)hipGraphExecDestroy(kv.second);
    a->graphs.clear();
    if (a->d_smallNeed) { (void)hipFree(a->d_smallNeed); a->d_smallNeed = nullptr; }
    for (auto& kv : a->gsTasks)
    {
        if (kv.second.d_tasks) (void)hipFree(kv.second.d_tasks);
        if (kv.second.d_slabTasks) (void)hipFree(kv.second.d_slabTasks);
    }
    a->gsTasks.clear();
    for (auto& kv : a->wgTasks)
        if (kv.second.d_tasks) (void)hipFree(kv.second.d_tasks);
    a->wgTasks.clear();
    void* ptrs[] = {a->d_perm, a->d_iperm, a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_sliceW, a->d_sliceT,
                    a->d_levelSliceStart, a->d_nL, a->d_nU, a->d_col, a->d_face, a->d_l, a->d_u,
                    a->d_losort, a->d_ownerStart, a->d_losortStart, a->d_bRow, a->d_bStart, a->d_bFace,
                    a->d_pfCell, a->d_sendAll, a->d_recvAll, a->p2p[0].d_granule, a->p2p[0].d_ticket,
                    a->p2p[1].d_granule, a->p2p[1].d_ticket, a->d_gateF, a->d_gateB,
                    a->d_sliceDone, a->d_nbRowB, a->d_nbK0, a->d_slabList, a->d_colX, a->d_xflag, a->p2p[0].d_X, a->p2p[0].d_ctl,
                    a->p2p[1].d_X, a->p2p[1].d_ctl};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (double* p : a->scratch) if (p) (void)hipFree(p);
    a->scratch.clear();
}

// Host-only plan statistics of an addressing (no device, no context): dependency levels of the lower-triangular
// DAG and the greedy cluster partition the cluster sweep engine would use.  For studying numberings and meshes on
// the build host.  out[0] dependency levels, [1] clusters, [2] cluster levels, [3] sum of the clusters' internal
// steps, [4] most internal steps, [5] most lower, [6] most upper neighbours of a row, [7] rows with more than 6
// lower or upper neighbours, [8] ... more than 12, [9] faces between different clusters, [10] widest dependency
// level (cells), [11] widest cluster level (clusters).
extern "C" int ldu_debug_dag_stats(int32_t nCells, int32_t nFaces, const int32_t* l, const int32_t* u, int32_t maxCells,
                                   int64_t out[16], int32_t* cellLevel, int32_t* cellCluster, int32_t* clusterLevel)
{
    if (nCells < 0 || nFaces < 0 || (nFaces && (!l || !u)) || !out) { ldu_set_error("ldu_debug_dag_stats: bad argument"); return -1; }
    const int nC = nCells, nF = nFaces;
    for (int f = 0; f < nF; f++)
        if (l[f] < 0 || u[f] >= nC || l[f] >= u[f] || (f && l[f] < l[f - 1]))
        {
            ldu_set_error("ldu_debug_dag_stats: faces must be in upper-triangular order");
            return -2;
        }
    std::vector<int> losort(nF), ownerStart(nC + 1, 0), losortStart(nC + 1, 0), level(nC, 0);
    {
        std::vector<int> cnt(nC + 1, 0);
        for (int f = 0; f < nF; f++) cnt[u[f] + 1]++;
        for (int c = 0; c < nC; c++) cnt[c + 1] += cnt[c];
        losortStart = cnt;
        for (int f = 0; f < nF; f++) losort[cnt[u[f]]++] = f;
        for (int f = 0; f < nF; f++) ownerStart[l[f] + 1]++;
        for (int c = 0; c < nC; c++) ownerStart[c + 1] += ownerStart[c];
    }
    for (int f = 0; f < nF; f++) level[u[f]] = std::max(level[u[f]], level[l[f]] + 1);
    int nLevels = 0;
    for (int c = 0; c < nC; c++) nLevels = std::max(nLevels, level[c] + 1);
    ClGreedy G;
    cluster_greedy(nC, nF, l, u, losort.data(), losortStart.data(), ownerStart.data(), level.data(),
                   maxCells > 0 ? maxCells : LDU_WAVE, G);
    for (int i = 0; i < 16; i++) out[i] = 0;
    out[0] = nLevels;
    out[1] = (int64_t)G.nClusters();
    int maxCl = 0;
    for (size_t i = 0; i < G.nClusters(); i++)
    {
        maxCl = std::max(maxCl, G.cLevel[i]);
        out[3] += G.cDepth[i];
        out[4] = std::max<int64_t>(out[4], G.cDepth[i]);
    }
    out[2] = (G.nClusters() == 0) ? 0 : maxCl + 1;
    for (int c = 0; c < nC; c++)
    {
        const int nl = losortStart[c + 1] - losortStart[c], nu = ownerStart[c + 1] - ownerStart[c];
        out[5] = std::max<int64_t>(out[5], nl);
        out[6] = std::max<int64_t>(out[6], nu);
        if (nl > 6 || nu > 6) out[7]++;
        if (nl > 12 || nu > 12) out[8]++;
    }
    for (int f = 0; f < nF; f++) if (G.cluster[l[f]] != G.cluster[u[f]]) out[9]++;
    {
        std::vector<int> w(nLevels + 1, 0), wc(out[2] + 1, 0);
        for (int c = 0; c < nC; c++) w[level[c]]++;
        for (size_t i = 0; i < G.nClusters(); i++) wc[G.cLevel[i]]++;
        for (int x : w) out[10] = std::max<int64_t>(out[10], x);
        for (int x : wc) out[11] = std::max<int64_t>(out[11], x);
    }
    if (cellLevel) for (int c = 0; c < nC; c++) cellLevel[c] = level[c];
    if (cellCluster) for (int c = 0; c < nC; c++) cellCluster[c] = G.cluster[c];
    if (clusterLevel) for (size_t i = 0; i < G.nClusters(); i++) clusterLevel[i] = G.cLevel[i];
    return 0;
}
