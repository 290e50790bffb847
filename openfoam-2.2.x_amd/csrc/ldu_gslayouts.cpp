// Per-sweep layouts of the chip-wide pipelined GaussSeidel sweeps (sweep_p2p_gs_multi_kernel, ldu_kernels.hip).
//
// k consecutive sweeps of GaussSeidelSmoother.C:147-176 in one launch: sweep j of row r needs sweep j's values of r's lower
// neighbours and sweep j-1's values of its upper neighbours.  In the row-level DAG of the k sweeps the earliest time of
// (j, r) is  T_0(r) = dependency level,  T_j(r) = 1 + max(T_j(lower neighbours), T_j-1(upper neighbours), T_j-1(r)),  and on the
// real motorBike mesh max T_j grows by 20-40 per sweep (GAMG level of 3.1 M cells: 669 / 692 / 719 for three sweeps).  The level
// layout cannot run at that pace: its slices hold the rows of ONE dependency level, a wave waits for all 64 rows, and the rows of
// a level need their upper neighbours - which sit anywhere up to hundreds of levels ahead - at very different times (the same
// level: 1939 steps of the slice DAG for the three sweeps, 1304 for two).  Lag buckets (ldu_plan.cpp) soften that inside one
// layout; here sweep j >= 1 gets its OWN slices - rows of equal T_j (and one width class) - with its own entry tables: col / face /
// nL / nU in slice order, the row itself through rowIdx (slot -> row of the level numbering; granules, psi, rhs, diag stay where
// they are).  A task (j, slice) then runs at time T_j exactly, any order by T is topological, and k sweeps take max T_k-1 steps.
// The arithmetic of a row is untouched (same entries, same order): bit-identical to the level layout and to k separate sweeps.
// (ldu_blocks.hip does the same inside LDS-resident blocks; this is the variant for levels that do not fit there.)
//
// MEASURED (mb12, profiles/r05_gs_layouts_probe.log) and therefore OFF by default (LDU_GS_LAYOUTS=1 turns it on): the levels that
// would take it are not bound by the depth of the slice DAG.  Finest level (12.7 M cells, 2 sweeps): 447 instead of 575 steps,
// 3.16 ms against 3.04; level of 6.3 M cells, 3 sweeps: 520 instead of 1400 steps, 4.90 ms against 4.04.  The task timeline of the
// finest level (tools/mesh_probe.py, PROBE_TRACE=1) says why: its 160 widest dependency levels hold 1300-2300 slices each - as many
// as the chip has wavefronts in flight - and take 10 us each because a task spends 4.5 us loading its tables and old values before
// it polls, 3.2 us polling; the narrow levels before and behind run at the hand-off latency (2.9-4.1 us per level).  Neither cost
// depends on how the rows of a later sweep are grouped, and a per-sweep layout adds a dependent gather (rowIdx -> rhs, diag) to
// every task.  Kept as a tested option: on DAGs whose pipelined sweeps ARE depth-bound on this engine it is the right tool.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include "ldu_internal.hpp"

// (row classes of ldu_plan.cpp)
static inline int gl_width_class(int cl, int cu)
{
    const int t = cl + cu;
    if (cl <= 8 && cu <= 8) return t <= 6 ? 0 : (t <= 8 ? 1 : (t <= 12 ? 2 : 3));
    return t <= 16 ? 4 : (t <= 32 ? 5 : (t <= 64 ? 6 : 7));
}
static inline int gl_class_lanes(int cls) { return cls == 4 ? 2 : (cls == 5 ? 4 : (cls == 6 ? 8 : 1)); }

template <class T>
static int gl_upload(T** dst, const std::vector<T>& src)
{
    const size_t n = std::max<size_t>(src.size(), 1);
    LDU_CHECK_HIP(hipMalloc((void**)dst, sizeof(T) * n));
    if (!src.empty()) LDU_CHECK_HIP(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    return 0;
}

static void gl_par(long n, long grain, const std::function<void(long, long)>& fn)
{
    const long nT = std::max<long>(1, std::min<long>(8, n / std::max<long>(1, grain)));
    if (nT <= 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    for (long t = 0; t < nT; t++) th.emplace_back(fn, n * t / nT, n * (t + 1) / nT);
    for (auto& x : th) x.join();
}

static void gl_free_one(ldu_addr::GsLayout* Y)
{
    if (!Y) return;
    void* ptrs[] = {Y->d_sliceRow, Y->d_sliceCnt, Y->d_sliceEnt, Y->d_sliceW, Y->d_sliceT, Y->d_nL, Y->d_nU, Y->d_col, Y->d_face, Y->d_rowIdx};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    delete Y;
}

void gs_layouts_free(ldu_addr* a)
{
    for (int j = 0; j < 4; j++) { gl_free_one(a->gsLay[j]); a->gsLay[j] = nullptr; }
    for (auto& kv : a->gsLayVals) for (int j = 0; j < 4; j++) if (kv.second.d[j]) (void)hipFree(kv.second.d[j]);
    a->gsLayVals.clear();
    a->gsLayState = 0;
    a->gsLayBuilt = 0;
}

void gs_layouts_forget(ldu_addr* a, const double* levelVal)
{
    auto it = a->gsLayVals.find(levelVal);
    if (it == a->gsLayVals.end()) return;
    for (int j = 0; j < 4; j++) if (it->second.d[j]) (void)hipFree(it->second.d[j]);
    a->gsLayVals.erase(it);
}

// does this addressing take per-sweep layouts?  (the caller has already decided that k sweeps run on the chip-wide engine)
bool gs_layouts_wanted(const ldu_addr* a)
{
    const ldu_ctx* ctx = a->ctx;
    return ctx->gsLayouts && ctx->sweepP2P && ctx->gsPipeline && !a->nPatchFaces && !a->wgLevel && a->nCells >= ctx->gsLayoutsMinCells
           && !ctx->p2pGate && a->gsLayState >= 0;
}

// layouts of sweeps 1 .. k-1; 0 = there, 1 = not on this addressing, < 0 error
int gs_layouts_ensure(ldu_addr* a, int k)
{
    if (k < 2 || k > 4 || !gs_layouts_wanted(a)) return 1;
    if (a->gsLayState == 1 && a->gsLayBuilt >= k) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    const int nC = a->nCells, nF = a->nFaces;
    const std::vector<int>& l = a->l;
    const std::vector<int>& u = a->u;
    const ldu_ctx* ctx = a->ctx;
    const bool coop = ctx->sortRowsByWidth && ctx->coopRows;
    // row times in the ORIGINAL numbering (lower neighbours have lower indices: one ascending pass per sweep)
    std::vector<std::vector<int>> T(k);
    T[0] = a->level;
    for (int j = 1; j < k; j++)
    {
        const std::vector<int>& Tp = T[j - 1];
        std::vector<int>& Tj = T[j];
        Tj = Tp;
        for (int f = 0; f < nF; f++) if (Tp[u[f]] > Tj[l[f]]) Tj[l[f]] = Tp[u[f]];
        for (int c = 0; c < nC; c++)
        {
            int t = Tj[c];
            for (int q = a->losortStart[c]; q < a->losortStart[c + 1]; q++) t = std::max(t, Tj[l[a->losort[q]]]);
            Tj[c] = t + 1;
        }
    }
    std::vector<unsigned char> cls(nC);
    for (int c = 0; c < nC; c++) cls[c] = (unsigned char)gl_width_class(a->losortStart[c + 1] - a->losortStart[c], a->ownerStart[c + 1] - a->ownerStart[c]);
    for (int j = std::max(1, a->gsLayBuilt); j < k; j++)
    {
        if (a->gsLay[j]) continue;
        const std::vector<int>& Tj = T[j];
        int maxT = 0;
        for (int c = 0; c < nC; c++) maxT = std::max(maxT, Tj[c]);
        // rows by (T_j, width class), original order inside
        std::vector<long> start((size_t)(maxT + 1) * 8 + 1, 0);
        for (int c = 0; c < nC; c++) start[(size_t)Tj[c] * 8 + cls[c] + 1]++;
        for (size_t i = 0; i + 1 < start.size(); i++) start[i + 1] += start[i];
        std::vector<int> order(nC);
        {
            std::vector<long> pos(start.begin(), start.end() - 1);
            for (int c = 0; c < nC; c++) order[(size_t)pos[(size_t)Tj[c] * 8 + cls[c]]++] = c;
        }
        ldu_addr::GsLayout* Y = new ldu_addr::GsLayout();
        std::vector<int> sliceRow, sliceCnt, sliceEnt, sliceW;
        std::vector<unsigned char> sliceT;
        std::vector<unsigned char> nL(nC), nU(nC);
        std::vector<int> rowIdx(nC);
        for (int i = 0; i < nC; i++)
        {
            const int c = order[i];
            nL[i] = (unsigned char)(a->losortStart[c + 1] - a->losortStart[c]);
            nU[i] = (unsigned char)(a->ownerStart[c + 1] - a->ownerStart[c]);
            rowIdx[i] = a->iperm[c];
        }
        long ent = 0;
        for (size_t key = 0; key + 1 < start.size(); key++)
        {
            const int cl = (int)(key & 7);
            const int Tl = coop ? gl_class_lanes(cl) : 1;
            const int maxCnt = LDU_WAVE / Tl;
            for (long r0 = start[key]; r0 < start[key + 1]; r0 += maxCnt)
            {
                const int cnt = (int)std::min<long>(maxCnt, start[key + 1] - r0);
                int W = 0;
                for (int i = 0; i < cnt; i++) W = std::max(W, (int)nL[r0 + i] + (int)nU[r0 + i]);
                sliceRow.push_back((int)r0);
                sliceCnt.push_back(cnt);
                sliceEnt.push_back((int)ent);
                sliceW.push_back(W);
                sliceT.push_back((unsigned char)Tl);
                if (Tl > 1) Y->coop = true;
                Y->sliceTime.push_back((int)(key >> 3));
                ent += (long)W * LDU_WAVE;
                if (ent > 2000000000L)
                {
                    delete Y;
                    a->gsLayState = -1;       // (entry offsets are 32-bit: this level keeps the level layout)
                    return 1;
                }
            }
        }
        Y->nSlices = (int)sliceRow.size();
        const long entPad = ent + 1024;       // (the fast path reads eight entries per row unconditionally)
        Y->nEntries = entPad;
        sliceRow.push_back(nC);
        std::vector<int> col((size_t)entPad, 0), face((size_t)entPad, -1);
        gl_par(Y->nSlices, 4096, [&](long s0, long s1) {
            for (long s = s0; s < s1; s++)
            {
                for (int i = 0; i < sliceCnt[s]; i++)
                {
                    const int slot = sliceRow[s] + i;
                    const int c = order[slot];
                    const long base = (long)sliceEnt[s] + i;
                    int q = 0;
                    for (int e = a->losortStart[c]; e < a->losortStart[c + 1]; e++, q++)
                    {
                        const int f = a->losort[e];
                        col[base + (long)q * LDU_WAVE] = a->iperm[l[f]];
                        face[base + (long)q * LDU_WAVE] = f;
                    }
                    for (int f = a->ownerStart[c]; f < a->ownerStart[c + 1]; f++, q++)
                    {
                        col[base + (long)q * LDU_WAVE] = a->iperm[u[f]];
                        face[base + (long)q * LDU_WAVE] = f;
                    }
                    for (; q < sliceW[s]; q++) col[base + (long)q * LDU_WAVE] = rowIdx[slot];     // padding: the row itself
                }
                for (int i = sliceCnt[s]; i < LDU_WAVE; i++)
                    for (int q = 0; q < sliceW[s]; q++) col[(long)sliceEnt[s] + i + (long)q * LDU_WAVE] = rowIdx[sliceRow[s]];
            }
        });
        // (nL / nU / rowIdx are read at slot + lane for every lane of a wave: pad to a full slice past the end)
        nL.resize((size_t)nC + LDU_WAVE, 0); nU.resize((size_t)nC + LDU_WAVE, 0); rowIdx.resize((size_t)nC + LDU_WAVE, 0);
        if (gl_upload(&Y->d_sliceRow, sliceRow) || gl_upload(&Y->d_sliceCnt, sliceCnt) || gl_upload(&Y->d_sliceEnt, sliceEnt)
            || gl_upload(&Y->d_sliceW, sliceW) || gl_upload(&Y->d_sliceT, sliceT) || gl_upload(&Y->d_nL, nL) || gl_upload(&Y->d_nU, nU)
            || gl_upload(&Y->d_col, col) || gl_upload(&Y->d_face, face) || gl_upload(&Y->d_rowIdx, rowIdx))
        {
            gl_free_one(Y);
            return -1;
        }
        a->gsLay[j] = Y;
        if (getenv("LDU_VERBOSE"))
            fprintf(stderr, "[ldugpu] GS layout of sweep %d: %d cells, %d slices (level layout: %d), %ld entries (%ld), row-DAG time %d "
                            "(one sweep: %d levels)\n", j, nC, Y->nSlices, a->nSlices, Y->nEntries, a->nEntries, maxT + 1, a->nLevels);
    }
    a->gsLayBuilt = std::max(a->gsLayBuilt, k);
    a->gsLayState = 1;
    if (getenv("LDU_VERBOSE"))
        fprintf(stderr, "[ldugpu] GS layouts up to k = %d of %d cells built in %.3f s\n", k, nC,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}

int k_fill_layout(const ldu_addr::GsLayout* Y, const double* lowerO, const double* upperO, double* val, hipStream_t s);   // ldu_kernels.hip

// the value arrays of layouts 1 .. k-1 for the level value array `levelVal` (filled from the face-ordered coefficients it was
// filled from; refilled when the coefficients changed: ctx->valStamp); 1 = origin unknown (the caller keeps the level layout)
int gs_layout_values(ldu_addr* a, const double* levelVal, int k, hipStream_t s, const double* out[4])
{
    auto org = a->valOrigin.find(levelVal);
    if (org == a->valOrigin.end()) return 1;
    ldu_addr::GsLayVals& V = a->gsLayVals[levelVal];
    out[0] = levelVal;
    for (int j = 1; j < 4; j++) out[j] = nullptr;
    for (int j = 1; j < k; j++)
    {
        const ldu_addr::GsLayout* Y = a->gsLay[j];
        if (!Y) return 1;
        if (!V.d[j]) LDU_CHECK_HIP(hipMalloc((void**)&V.d[j], sizeof(double) * (size_t)Y->nEntries));
        if (V.stamp[j] != val_stamp(a, levelVal))
        {
            if (k_fill_layout(Y, org->second.first, org->second.second, V.d[j], s)) return -1;
            V.stamp[j] = val_stamp(a, levelVal);
        }
        out[j] = V.d[j];
    }
    return 0;
}
