// Block engine: k pipelined GaussSeidel sweeps of a mid-size matrix (GAMG levels of ~6 000 ... ~2.5 M cells on unstructured
// meshes) with the dependency hand-offs INSIDE a workgroup's LDS.
//
// Why.  The sequential GaussSeidel recurrence (GaussSeidelSmoother.C:147-176) is executed in the reference's order, row by
// row, so a sweep costs (depth of the dependency DAG) x (time of one hand-off).  The chip-wide / slab engines hand over
// through L2 (2.5-3 us per dependency level on the agglomerated levels of the motorBike mesh: 12-15 neighbours per row,
// 400-900 levels deep); the one-workgroup engine (gs_wg_kernel) hands over through LDS but holds the whole matrix in ONE
// workgroup (<= 6 000 cells).  Here the matrix is cut into compact BLOCKS of a few thousand cells; every block is one
// workgroup that keeps the solution values AND a sweep stamp per row of its block in LDS and runs the k sweeps as
// (sweep, group of rows) tasks of its wavefronts.  Only dependencies that cross a block boundary travel through memory: a row
// with a neighbour in another block publishes {value, tag} as a 16-byte granule (the engines' common hand-off format); the
// consuming block has a GHOST slot for that row in its LDS, filled by a dedicated importer wavefront that polls the
// granules in the order in which they become due.
//
// Groups.  A task is a group of rows of one block that can run at the same time.  For sweep j the rows are grouped by their
// time T_j in the ROW-level dependency DAG of the k sweeps (T_0 = dependency level; T_j(r) = 1 + max(T_j of the lower
// neighbours, T_j-1 of the upper neighbours, T_j-1(r))): every sweep has its own grouping and its own copy of the row and
// entry tables.  Grouping every sweep by the dependency level (what the level engines' slices do) makes sweep j + 1 trail
// sweep j by the WORST upper-neighbour reach inside a group - on the agglomerated levels four sweeps then cost 3.7 x one
// (1247 steps against 518 on the 93 k-cell level of the 12.7 M-cell motorBike mesh, tools/block_probe2.cpp).
// Rows of more than 16 entries are spread over 2 / 4 / 8 lanes (16 entries per lane in registers, the partial sums handed
// from lane to lane in entry order), so a wide row costs no memory round trip on the dependency chain.
//
// Exactness.  Per row: acc = b; acc -= coeff * value in the reference's accumulation order (lower-neighbour faces ascending,
// then owned faces ascending); ldu_div - the arithmetic of every other engine, bit-identical to the sequential loop.
// The readiness rule is the exact dependency of that loop: a row of sweep j needs its lower neighbours with stamp j + 1 and
// its upper neighbours with stamp j; a stamp can never be AHEAD of what a row needs (the neighbour's next sweep needs this
// row's current one), so one value slot per row suffices - in LDS and in the granule array.
//
// Progress.  Within a sweep a group holds rows of ONE value of a strict potential of the row DAG, so the (sweep, group)
// tasks of all blocks form an acyclic graph; Phi (the group-level version of T) orders them.  Every worker - a compute
// wavefront's task list, a lane of the importer - walks its items in non-decreasing Phi: the unfinished item of smallest
// Phi has all its inputs and is its worker's current item - PROVIDED its block has been started.  The quotient graph of
// the blocks is cyclic, so a started block may wait for one that is not.  Workgroups therefore take their block from a
// ticket counter, in the order of the blocks' smallest Phi: the started blocks are always a prefix of that order.  Let t*
// be the unfinished task of smallest Phi and b its block: every started, unfinished block before b has a task at or below
// Phi(t*) and one at or above it - it is OPEN at Phi(t*), and so is b.  If the chip holds at least as many workgroups as
// blocks are ever open at one value of Phi (openMax, computed from the task lists), b cannot be waiting for a slot: b is
// started, t* runs.  Levels whose blocks are all resident at once are the special case openMax <= blocks <= capacity;
// the 6.3 M- and 3.1 M-cell levels of the motorBike hierarchy run with 785 / 390 blocks of which ~150 / ~100 are open at a
// time.  Nothing is assumed about the order in which the hardware starts workgroups.  Every wait is bounded like in the
// other engines (abort flag -> the operation is re-run on the level kernels): a launch fails loudly, it never hangs.
// psi is read at block start and ghosts of late blocks need the OLD values of rows whose block is long done: the sweeps
// write their result to a scratch vector, copied over psi behind the launch.
//
// Coupled interfaces (cyclic patches: sub-domain mode, ldu_addr_set_subdomains).  GaussSeidelSmoother.C:98-145 adds the
// interface terms to bPrime before every sweep, with the neighbour sub-domain's values of the PREVIOUS sweep and negated
// coefficients, patch by patch, face by face.  Here they are the first entries of a row (coefficient -bouCoeffs, the
// neighbour cell across the face), ahead of the lower and upper entries: the same subtractions in the same order.  Both
// sides of an interface want each other's OLD value, so "a stamp is never ahead of what a row needs" does not hold
// across it: a row can finish sweep j before its interface neighbour has read its value of sweep j - 1.  One sweep
// ahead is all it can get (its sweep j + 1 needs the neighbour's sweep j, which has read), so interface cells keep TWO
// values, by sweep parity - in LDS ("hist" slots, for local cells and for ghosts) and in the granule array.
//
// Remote interfaces (processor patches, peer-store backend: ranks of one node).  The cell across a processor-patch face lives
// on another rank; here it is a MIRROR: a cell label >= nCells (nCells + patch-face index) that belongs to no block and has no
// row - only a hist pair in the LDS of the blocks that reference it, filled by the importer like any ghost across an interface.
// What fills it comes through the peer window: the rank across the face stores its interface cells' values - the initial one
// when its block starts, then the value after every sweep but the last - into THIS rank's window with one system-scope store
// per face (ldu_peer_dev.hpp: peer_store), tagged bSeq + 1 + sweeps-done, parity = tag & 1 (PeerHalo::d_bdst / d_bsrc, a region
// of their own: the halo exchanges and the one-launch kernels of small levels keep theirs).  The tag sequence is counted on
// the host: every rank launches the same smoothing calls on a level, k exchanges per launch of k sweeps.  A face's slot of
// parity p is rewritten two exchanges later, after the writer's own row across that face has used the value the reader
// produced from it - the double buffering of ldu_peer.hip, face by face.  Whether a level runs this way is decided
// collectively (k_blocks_peer_decide: every rank's own plan must exist), because the ranks across its interfaces must speak the
// same protocol; every wait of such a launch is bounded by the peer time-out (a neighbour may simply be late).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <thread>
#include <vector>

#include "ldu_internal.hpp"
#include "ldu_peer_dev.hpp"

#define BK_MAX_LDS (160 * 1024)
#define BK_NLAY 4

typedef unsigned int bk_u32x4 __attribute__((ext_vector_type(4)));

struct BlockPlan {
    bool built = false;
    bool eligible = false;
    int nBlocks = 0;
    int maxSlots = 0;              // max over blocks of local rows + ghosts
    int nGhostTotal = 0;
    int nw = 7;                    // compute wavefronts per block (+ 1 importer): 7 = one workgroup of 512 threads per CU, 3 = two of 256
    int nLayouts = BK_NLAY;        // sweep j uses the grouping of layout min(j, nLayouts - 1)
    size_t ldsBytes = 0;
    // device tables
    int4* d_blk = nullptr;         // [nBlocks] {first local row in d_localRow, nLocal, ghostBase, nGhost}
    int* d_localRow = nullptr;     // [nCells] slot order of every block: level-ordered row
    int* d_ghostRow = nullptr;     // [nGhostTotal] level-ordered row of a ghost
    struct Layout {
        int nGroups = 0;
        long nLanes = 0, nEntries = 0;
        int4* d_meta = nullptr;            // [nLanes + 64] {level-ordered row | exported << 31 | last lane of the row << 30,
                                           //  LDS slot | lower entries of this lane << 16 | entries << 21 | interface entries << 26,
                                           //  first LDS slot of the row's hist pair (-1: no interface), index of its interface granules}
        unsigned* d_col = nullptr;         // [nEntries / 2] LDS slots of the columns of entries 2 p, 2 p + 1 (low, high half)
        int* d_srcFace = nullptr;          // [nEntries] face << 1 | (1: upper-triangle coefficient), -1 padding
        // host
        std::vector<int> grpBlk, grpLane0, grpCnt, grpT, grpEnt, grpStride, Phi, grpOfCell;
    } lay[BK_NLAY];
    uint4* d_granule = nullptr;    // [nCells + 1]
    unsigned* d_ticket = nullptr;  // block tickets (monotonic: a launch takes nBlocks of them)
    unsigned ticketBase = 0;
    double* d_out = nullptr;       // [nCells] result of a launch before it is copied over psi
    long capacity = 0;             // workgroups of the kernel the chip holds at once
    unsigned epoch = 0;
    int gen = 0;
    struct Tasks { int4* d_tasks = nullptr; int* d_taskStart = nullptr; int4* d_imps = nullptr; int* d_impStart = nullptr; long nTasks = 0;
                   int* d_order = nullptr; int openMax = 0; bool usable = true; };
    std::map<int, Tasks> tasks;    // per k
    struct Conv { double* d[BK_NLAY] = {nullptr, nullptr, nullptr, nullptr}; unsigned long long stamp[BK_NLAY] = {0, 0, 0, 0}; };
    std::map<const double*, Conv> conv;
    // host copies the per-k lists are made from
    std::vector<int> ghostBase;            // [nBlocks + 1]
    std::vector<int> ghostCell;            // [nGhostTotal] old cell label
    std::vector<unsigned char> ghostLower; // [nGhostTotal] 1 = lower neighbour of a local row (needs stamps 1 ... k)
    std::vector<int> ghostRowH;            // [nGhostTotal] level-ordered row
    std::vector<int> blkNLocal;            // [nBlocks]
    std::vector<int> blk, slot, rowBase;   // [nCells] block / LDS slot of a cell (old labels), [nBlocks + 1]
    std::vector<char> exported;            // [nCells] has a neighbour in another block
    std::vector<int> RTlast;               // T of the last grouping built
    int nBuilt = 0;                        // groupings built so far (sweeps 0 ... nBuilt - 1)
    // coupled (cyclic) interfaces
    bool iface = false;
    int nIf = 0;                           // cells with interface faces
    std::vector<int> ifStart, ifPf, ifNbr; // [nCells + 1] CSR: a cell's patch faces in (patch, face) order: patch-face index, neighbour cell
    std::vector<int> ifIdx;                // [nCells] index among the interface cells, -1
    std::vector<int> histBase, histCell;   // [nBlocks + 1]; cells with a hist pair per block: its own interface cells, then interface ghosts
    int2* d_blk2 = nullptr;                // [nBlocks] {histBase, nHist}
    int* d_histRow = nullptr;              // [histCell.size()] level-ordered row; a mirror: -1 - patch face
    // remote interfaces (processor patches; see "Remote interfaces" at the top)
    bool remote = false;
    std::vector<int> pfCell;               // [nPatchFaces] the local cell of a patch face (old label)
    // [layout][nPatchFaces] group-level time Phi of the cell ACROSS a processor-patch face in that layout (exchanged when the
    // plan is made, k_blocks_peer_decide): the Phi of this rank's groups are potentials of the task graph of ALL ranks - a
    // worker walks its tasks in ascending Phi, and a task that waits for another rank must not stand in front of one that rank
    // is waiting for
    std::vector<int> remPhi[BK_NLAY];
    int* d_remStart = nullptr;             // [nIf + 1] CSR: the processor-patch faces of an interface cell
    int* d_remPf = nullptr;
};

template <class T>
static int bk_upload(T** dst, const std::vector<T>& src, size_t extra = 0)
{
    const size_t n = src.size() + extra ? src.size() + extra : 1;
    LDU_CHECK_HIP(hipMalloc((void**)dst, n * sizeof(T)));
    if (extra || !src.size()) LDU_CHECK_HIP(ldu_memset_sync(*dst, 0, n * sizeof(T)));
    if (src.size()) LDU_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

void blocks_free(ldu_addr* a)
{
    BlockPlan* P = a->blocks;
    if (!P) return;
    void* ptrs[] = {P->d_blk, P->d_localRow, P->d_ghostRow, P->d_granule, P->d_ticket, P->d_out, P->d_blk2, P->d_histRow,
                    P->d_remStart, P->d_remPf};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& L : P->lay)
        for (void* p : {(void*)L.d_meta, (void*)L.d_col, (void*)L.d_srcFace}) if (p) (void)hipFree(p);
    for (auto& kv : P->conv) for (double* d : kv.second.d) if (d) (void)hipFree(d);
    for (auto& kv : P->tasks)
        for (void* p : {(void*)kv.second.d_tasks, (void*)kv.second.d_taskStart, (void*)kv.second.d_imps, (void*)kv.second.d_impStart, (void*)kv.second.d_order})
            if (p) (void)hipFree(p);
    delete P;
    a->blocks = nullptr;
}

void blocks_forget(ldu_addr* a, const double* levelVal)
{
    if (!a) return;
    addr_bg_wait(a);      // (a plan thread may be filling a->blocks)
    if (!a->blocks) return;
    auto it = a->blocks->conv.find(levelVal);
    if (it == a->blocks->conv.end()) return;
    (void)hipStreamSynchronize(a->ctx->stream);
    for (double* d : it->second.d) if (d) (void)hipFree(d);
    a->blocks->conv.erase(it);
}

// ---------------------------------------------------------------- kernels

static __device__ unsigned long long* g_bk_trace = nullptr;

int k_blocks_set_watchdog(unsigned long long budgetTicks, unsigned long long stallTicks)
{
    unsigned long long v[2] = {budgetTicks, stallTicks};
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wait_budget), v, sizeof(v)));
    return 0;
}

__global__ void __launch_bounds__(256)
bk_fill_kernel(long n, const int* __restrict__ srcFace, const double* __restrict__ lowerO, const double* __restrict__ upperO,
               const double* __restrict__ bou, double* __restrict__ out)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    {
        const int code = srcFace[i];
        double v = 0.0;
        if (code >= 0) v = (code & 1) ? upperO[code >> 1] : lowerO[code >> 1];
        else if (code < -1) v = -bou[-2 - code];     // interface face: the negated interfaceBouCoeffs (GaussSeidelSmoother.C:98-107)
        out[i] = v;
    }
}

__device__ __forceinline__ void bk_store(uint4* G, int row, double v, unsigned tag)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    bk_u32x4 d;
    d.x = (unsigned)b; d.y = tag; d.z = (unsigned)(b >> 32); d.w = tag;
    uint4* p = G + row;
    // (s_nop 1: the two wait states a VMEM store of more than 64 bits needs before its data registers are rewritten -
    //  the hazard recognizer does not look inside inline asm, DESIGN.md "A hardware hazard worth recording")
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}

// Stores of the rare paths (abort flag, trace stamps) as inline asm: a compiler-visible store anywhere in the step loop leaves
// "loads AND stores pending" on vmcnt (gfx9 counts both; they return out of order with respect to each other), and the
// waitcnt pass then turns every wait of the loop into s_waitcnt vmcnt(0) - the prefetches in flight included.
__device__ __forceinline__ void bk_flag_set(int* p)
{
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(1) : "memory");
}
__device__ __forceinline__ void bk_trace_store(unsigned long long* p, unsigned long long v)
{
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ bk_u32x4 bk_load(const uint4* p)
{
    bk_u32x4 g;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
    return g;
}

// the value of the previous lane (row_shr:1 inside a row of 16 lanes; groups of 2 / 4 / 8 lanes never straddle one)
__device__ __forceinline__ double bk_from_prev_lane(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x111, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

struct BkTab {
    const int4* blk; const int* localRow; const int* ghostRow; const int2* blk2; const int* histRow; int ifBase;
    const int4* meta[BK_NLAY]; const unsigned* col[BK_NLAY]; const double* val[BK_NLAY];
    int nLayouts;
    const int4* tasks; const int* taskStart; const int4* imps; const int* impStart;
    // remote interfaces (null / 0 without): destination and source granules per (parity, patch face) in the peer windows, the tag
    // of this launch's first exchange, its number of sweeps, the processor-patch faces of the interface cells
    uint4* const* bdst; const uint4* const* bsrc; int nPF; unsigned S0; int kSweeps;
    const int* remStart; const int* remPf;
};

// a dependency wait of a launch with remote interfaces is (also) a wait for another rank: the peer time-out, not the 200 ms
__device__ __forceinline__ bool bk_wait_expired(bool remote, unsigned& spins, int* abortFlag, unsigned long long& tw0)
{
    if (!remote) return ldu_wait_expired(spins, 1u << 30, abortFlag, tw0);
    if ((spins & 255u) == 7u && *(volatile int*)abortFlag) return true;      // (another wave of this rank gave up: drain)
    return peer_wait_expired(spins, tw0, abortFlag);
}

// a task's rows in flight: one lane = one row (T = 1) or one sixteen-entry part of a row (T = 2 / 4 / 8 lanes per row)
struct BkRow { int4 Q; int rg, slot, nl, nn, ni, hist, ifg, T, j; bool have; unsigned c2[8]; double v[16]; double b, d; };

// stage B of a task's prefetch: everything that depends on the task record (R.Q, loaded a step earlier) alone
__device__ __forceinline__ void bk_row_load(int lane, const BkTab& T, BkRow& R)
{
    // (the record was loaded a step ago into VECTOR registers - see BK_REC - and becomes scalar here, where it is first needed)
    const int4 Q = make_int4(__builtin_amdgcn_readfirstlane(R.Q.x), __builtin_amdgcn_readfirstlane(R.Q.y),
                             __builtin_amdgcn_readfirstlane(R.Q.z), __builtin_amdgcn_readfirstlane(R.Q.w));
    const int cnt = Q.y & 255, stride = (Q.y >> 16) & 255, j = Q.w;
    const int L = j < T.nLayouts ? j : T.nLayouts - 1;
    // (constant indices + scalar selects: T lives in the kernel's argument segment, and T.meta[L] with a run-time L is a scalar
    //  LOAD from it - s_load + s_waitcnt lgkmcnt(0) - in every step)
#define BK_SEL(A) (L == 0 ? (A)[0] : L == 1 ? (A)[1] : L == 2 ? (A)[2] : (A)[3])
    const int4* __restrict__ meta = BK_SEL(T.meta);
    const unsigned* __restrict__ col = BK_SEL(T.col);
    const double* __restrict__ val = BK_SEL(T.val);
#undef BK_SEL
    const bool have = lane < cnt;
    const int4 M = meta[Q.x + (have ? lane : 0)];
    const long ent2 = (long)(Q.z >> 1) + (have ? lane : 0);    // pairs of coefficients (16 bytes)
    const long ent8 = (long)(Q.z >> 3) + (have ? lane : 0);    // quads of column words (8 slots, 16 bytes)
    R.rg = M.x;
    R.slot = M.y & 0xffff;
    R.nl = (M.y >> 16) & 31;
    R.nn = (M.y >> 21) & 31;
    R.ni = (M.y >> 26) & 31;
    R.hist = M.z;
    R.ifg = M.w;
    const uint4* __restrict__ col4 = (const uint4*)col;
    const double2* __restrict__ val2 = (const double2*)val;
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
        const uint4 c = col4[ent8 + (long)r * stride];
        R.c2[4 * r] = c.x; R.c2[4 * r + 1] = c.y; R.c2[4 * r + 2] = c.z; R.c2[4 * r + 3] = c.w;
    }
#pragma unroll
    for (int p = 0; p < 8; p++)      // (a group's entries: 16 x stride, stride >= its lanes)
    {
        const double2 v = val2[ent2 + (long)p * stride];
        R.v[2 * p] = v.x; R.v[2 * p + 1] = v.y;
    }
    R.have = have;
    R.T = (Q.y >> 8) & 255;
    R.j = j;
}

template <int NW>
__global__ void __launch_bounds__(LDU_WAVE * (NW + 1))
gs_blk_kernel(BkTab T, int nBlocks, const int* __restrict__ order, uint4* __restrict__ G, unsigned tagBase, unsigned* ticketCtr,
              unsigned ticketBase, int* abortFlag, const double* __restrict__ psi, double* __restrict__ psiOut,
              const double* __restrict__ rhs, const double* __restrict__ diag)
{
    extern __shared__ double smem[];
    // the block of this workgroup: the next one in the order of the blocks' first tasks (see "Progress" above) - whatever
    // order the hardware starts workgroups in, the started blocks are a prefix of that order
    if (threadIdx.x == 0)
    {
        const unsigned t = atomicAdd(ticketCtr, 1u) - ticketBase;
        ((int*)smem)[0] = order[t];
        ((int*)smem)[1] = (int)t;
    }
    __syncthreads();
    const int b = ((const int*)smem)[0];
    const bool firstBlock = ((const int*)smem)[1] == 0;
    __syncthreads();      // (smem is about to become the block's value slots)
    const int4 B = T.blk[b];
    const int rowBase = B.x, nLocal = B.y, ghostBase = B.z, nGhost = B.w;
    const int histBase = T.blk2 ? T.blk2[b].x : 0, nHist = T.blk2 ? T.blk2[b].y : 0;
    const int nSlots = nLocal + nGhost + 2 * nHist;
    double* x = smem;
    unsigned char* stamp = (unsigned char*)(x + nSlots);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // every compute wavefront has its own task list (taskStart[b * NW + wave]): the tasks of ONE sweep - inside a block a sweep is
    // nearly a chain of dependent groups, and a wavefront sees its own LDS writes in program order: no hand-off between
    // wavefronts along the chain, only between the sweeps
    const int t0 = wave < NW ? T.taskStart[b * NW + wave] : 0;
    const int nTasks = wave < NW ? T.taskStart[b * NW + wave + 1] - t0 : 0;
    const int4* tasks = T.tasks + t0;
    BkRow R0, R1, R2;
    int iNext = 0;
    // a task's data arrive in three stages, each a step (= one task of this wavefront) ahead of the next: its record; its rows'
    // meta data, columns and coefficients (they depend on the record only); rhs and diagonal (they depend on the row).  The
    // record of a buffer's NEXT task is requested at the start of the step that computes its current one.
    // The record's address is wave-uniform, and the compiler reads a uniform load's result into scalar registers RIGHT BEHIND the
    // load (global_load; s_waitcnt vmcnt(0); v_readfirstlane): a full memory round trip - and a wait for every prefetch in
    // flight - at the start of every step, on the dependency chain (0.68 of the 1.1 us of a step in the round-5 trace).  An
    // index the compiler cannot prove uniform (vz = 0 from an opaque asm) keeps the record in vector registers until
    // bk_row_load makes it scalar a step later.
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
#define BK_REC(R) do { (R).Q = iNext < nTasks ? tasks[iNext + vz] : make_int4(0, 0, 0, 0); iNext++; } while (0)
#define BK_FILLB(R) bk_row_load(lane, T, (R))
#define BK_FILLC(R) do { const int g_ = (R).rg & 0x3fffffff; (R).b = rhs[g_]; (R).d = diag[g_]; } while (0)
    if (wave < NW)
    {
        BK_REC(R0);
        BK_REC(R1);
        BK_REC(R2);
        BK_FILLB(R0);
        BK_FILLB(R1);
    }
    for (int i = tid; i < nLocal; i += LDU_WAVE * (NW + 1)) { x[i] = psi[T.localRow[rowBase + i]]; stamp[i] = 0; }
    for (int i = tid; i < nGhost; i += LDU_WAVE * (NW + 1)) { x[nLocal + i] = psi[T.ghostRow[ghostBase + i]]; stamp[nLocal + i] = 0; }
    // hist pairs of the interface cells (own and ghosts): slot 0 = the initial value (stamp 0), slot 1 = what sweep 0 leaves
    const bool remote = T.bsrc != nullptr;
    for (int i = tid; i < nHist; i += LDU_WAVE * (NW + 1))
    {
        const int h = nLocal + nGhost + 2 * i;
        const int hr = T.histRow[histBase + i];
        // (a mirror - the cell across a processor-patch face -: nothing is there yet, stamp 255 matches no sweep; the
        //  importer brings the neighbour's initial value with stamp 0)
        x[h] = hr >= 0 ? psi[hr] : 0.0; x[h + 1] = 0.0;
        stamp[h] = hr >= 0 ? 0 : 255; stamp[h + 1] = 0;
    }
    __syncthreads();
    bool alive = true;
    if (wave == NW)
    {
        // ---- importer: lane L walks the block's import records L, L + 64, ... (sorted by the time they become due); a
        // record = {ghost slot, level-ordered row, stamp}; the granule of the row carries tag tagBase + stamp once the
        // producing block has finished that sweep of the row
        const int i0 = T.impStart[b], nImp = T.impStart[b + 1] - i0;
        const int4* I = T.imps + i0;
        int i = lane;
        int4 e = i < nImp ? I[i] : make_int4(0, 0, 0, 0);
        int4 en = i + LDU_WAVE < nImp ? I[i + LDU_WAVE] : make_int4(0, 0, 0, 0);
        // where a record's granule is and what tag it waits for: a row of another block of this rank (G, tagBase + stamp), or -
        // e.y < 0 - the mirror of processor-patch face -1 - e.y: the granule of this rank's window into which the rank across
        // the face stores (tag S0 + stamp, parity by tag)
        const uint4* gp = G;
        unsigned want = 0;
#define BK_IMP_PREP()                                                                                     \
        do {                                                                                              \
            if (e.y >= 0) { gp = G + e.y; want = tagBase + (unsigned)e.z; }                               \
            else { want = T.S0 + (unsigned)e.z; gp = T.bsrc[(size_t)(want & 1u) * T.nPF + (-1 - e.y)]; } \
        } while (0)
        if (i < nImp) BK_IMP_PREP();
        unsigned spins = 0;
        unsigned long long tw0 = 0;
        while (true)
        {
            const bool have = i < nImp;
            if (__builtin_amdgcn_ballot_w64(have) == 0ull) break;
            bool ok = false;
            if (have)
            {
                // (plans with remote interfaces poll everything with system-scope loads: one load flavour per iteration)
                bk_u32x4 g;
                if (remote) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(gp) : "memory");
                else g = bk_load(gp);
                ok = g.y == want && g.w == want;
                if (ok)
                {
                    x[e.x] = __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
                    LDU_LDS_RELEASE();
                    stamp[e.x] = (unsigned char)e.z;
                    i += LDU_WAVE;
                    e = en;
                    en = i + LDU_WAVE < nImp ? I[i + LDU_WAVE] : make_int4(0, 0, 0, 0);
                    if (i < nImp) BK_IMP_PREP();
                }
            }
            if (__builtin_amdgcn_ballot_w64(ok) != 0ull) { spins = 0; tw0 = 0; }
            else
            {
                __builtin_amdgcn_s_sleep(1);
                if (bk_wait_expired(remote, spins, abortFlag, tw0)) { bk_flag_set(abortFlag); break; }
            }
        }
    }
    else
    {
        BK_FILLC(R0);
        ldu_debug_stall(firstBlock && wave == 0);    // (tests: the first task of the launch sits still, ldu_ctx_set_watchdog)
        int left = nTasks;      // this wavefront's tasks
        unsigned long long* trc = g_bk_trace ? g_bk_trace + (size_t)t0 * 8 : nullptr;
#define BK_TRC(k) do { if (trc && lane == 0) bk_trace_store(trc + (k), wall_clock64()); } while (0)
#define BK_STEP(CUR, NXT, FILL)                                                           \
    do {                                                                                  \
        BK_TRC(0);                                                                        \
        BK_REC(CUR);                                                                      \
        BK_FILLC(NXT);                                                                    \
        BK_FILLB(FILL);                                                                   \
        BK_TRC(1);                                                                        \
        const double rd_ = ldu_div_prepare((CUR).d);                                      \
        {                                                                                 \
            const bool have = (CUR).have;                                                 \
            const int self = have ? (CUR).slot : 0;                                       \
            const int nl = (CUR).nl, nn = have ? (CUR).nn : 0, j = (CUR).j;               \
            int cc[16];                                                                   \
            const int ni = (CUR).ni;                                                      \
            _Pragma("unroll") for (int q = 0; q < 16; q++)                                \
                cc[q] = q < nn ? (int)((q & 1) ? (CUR).c2[q >> 1] >> 16 : (CUR).c2[q >> 1] & 0xffffu) + (q < ni ? (j & 1) : 0) : self; \
            /* lower entries need stamp j + 1, upper entries and the padding (the row itself) stamp j; a lane without a row \
               reads stamp[0] sixteen times over and does not vote */                     \
            const int want = 16 * j + nl;                                                 \
            double xv[16], pr[16];                                                        \
            {                                                                             \
                unsigned spins = 0;                                                       \
                unsigned long long tw0 = 0;                                               \
                while (alive)                                                             \
                {                                                                         \
                    int st[16];                                                           \
                    _Pragma("unroll") for (int q = 0; q < 16; q++) st[q] = stamp[cc[q]];  \
                    int sum = 0;                                                          \
                    _Pragma("unroll") for (int q = 0; q < 16; q++) sum += st[q];          \
                    if (__builtin_amdgcn_ballot_w64(have && sum != want) == 0ull) break;  \
                    if (bk_wait_expired(remote, spins, abortFlag, tw0)) { bk_flag_set(abortFlag); alive = false; } \
                }                                                                         \
                LDU_LDS_ACQUIRE();                                                        \
                _Pragma("unroll") for (int q = 0; q < 16; q++) xv[q] = x[cc[q]];          \
            }                                                                             \
            BK_TRC(2);                                                                    \
            _Pragma("unroll") for (int q = 0; q < 16; q++) asm volatile("" : "+v"(xv[q])); \
            _Pragma("unroll") for (int q = 0; q < 16; q++)                                \
                pr[q] = q < nn ? (CUR).v[q] * xv[q] : 0.0;                                \
            double acc = (CUR).b;                                                         \
            const int Tl = (CUR).T;                                                       \
            if (Tl == 1)                                                                  \
            {                                                                             \
                _Pragma("unroll") for (int q = 0; q < 16; q++) acc -= pr[q];              \
            }                                                                             \
            else                                                                          \
            {                                                                             \
                /* a row of more than 16 entries: lane t of its Tl lanes holds entries 16 t ... 16 t + 15; the partial sum \
                   travels from lane to lane, every lane subtracting its products in entry order */ \
                const int t = lane & (Tl - 1);                                            \
                for (int s_ = 0; s_ < Tl; s_++)                                           \
                {                                                                         \
                    const double up = bk_from_prev_lane(acc);                             \
                    double a2 = s_ ? up : acc;                                            \
                    _Pragma("unroll") for (int q = 0; q < 16; q++) a2 -= pr[q];           \
                    acc = t == s_ ? a2 : acc;                                             \
                }                                                                         \
            }                                                                             \
            if (have && ((CUR).rg & 0x40000000))                                          \
            {                                                                             \
                const double xn_ = ldu_div(acc, (CUR).d, rd_);                            \
                x[self] = xn_;                                                            \
                /* a row with a neighbour in another block publishes its granule (first: the longer way) */ \
                if ((CUR).rg < 0) bk_store(G, (CUR).rg & 0x3fffffff, xn_, tagBase + (unsigned)j + 1u); \
                const int hs_ = (CUR).hist;                                               \
                if (hs_ >= 0)                                                             \
                {                                                                         \
                    /* an interface cell: its value by sweep parity, for the rows across the interface (here and elsewhere) */ \
                    x[hs_ + ((j + 1) & 1)] = xn_;                                         \
                    bk_store(G, T.ifBase + 2 * (CUR).ifg + ((j + 1) & 1), xn_, tagBase + (unsigned)j + 1u); \
                    if (remote && j + 1 < T.kSweeps)                                      \
                    {                                                                     \
                        /* ... and for the ranks across its processor-patch faces (their next sweep reads it) */ \
                        const unsigned tg_ = T.S0 + (unsigned)j + 1u;                     \
                        for (int e_ = T.remStart[(CUR).ifg]; e_ < T.remStart[(CUR).ifg + 1]; e_++) \
                            peer_store(T.bdst[(size_t)(tg_ & 1u) * T.nPF + T.remPf[e_]], xn_, tg_); \
                    }                                                                     \
                }                                                                         \
                LDU_LDS_RELEASE();                                                        \
                stamp[self] = (unsigned char)(j + 1);                                     \
                if (hs_ >= 0) stamp[hs_ + ((j + 1) & 1)] = (unsigned char)(j + 1);        \
            }                                                                             \
        }                                                                                 \
        LDU_STEP_FENCE();                                                                 \
        if (trc && lane == 0) { bk_trace_store(trc + 3, wall_clock64()); bk_trace_store(trc + 4, wave); bk_trace_store(trc + 5, (CUR).j); \
                                bk_trace_store(trc + 6, b); bk_trace_store(trc + 7, (CUR).T); } \
        if (trc) trc += 8;                                                                \
        --left;                                                                           \
    } while (0)
        while (left > 0)
        {
            BK_STEP(R0, R1, R2);
            if (left == 0) break;
            BK_STEP(R1, R2, R0);
            if (left == 0) break;
            BK_STEP(R2, R0, R1);
        }
#undef BK_STEP
#undef BK_TRC
    }
#undef BK_FILLB
#undef BK_FILLC
#undef BK_REC
    __syncthreads();
    for (int i = tid; i < nLocal; i += LDU_WAVE * (NW + 1)) psiOut[T.localRow[rowBase + i]] = x[i];
}

// Remote interfaces: the INITIAL values of this rank's cells at its processor-patch faces, for the ranks across them (tag S0).
// A launch of its own in front of the block kernel: these values need no block to have started - a block of the other rank may
// wait for them long before the block that owns the cell here gets its turn.
__global__ void __launch_bounds__(256) bk_init_export_kernel(int nPF, const int* __restrict__ pfRow, const double* __restrict__ psi,
                                                             uint4* const* __restrict__ bdst, unsigned S0)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nPF; i += gridDim.x * 256)
    {
        uint4* d = bdst[(size_t)(S0 & 1u) * nPF + i];
        if (d) peer_store(d, psi[pfRow[i]], S0);
    }
}

__global__ void __launch_bounds__(256) bk_copy_kernel(long n, const double* __restrict__ src, double* __restrict__ dst)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}

// ---------------------------------------------------------------- host: the plan

template <int NW>
static int bk_occupancy(size_t lds, int* perCU)
{
    LDU_CHECK_HIP(hipFuncSetAttribute((const void*)gs_blk_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, BK_MAX_LDS));
    int n = 0;
    LDU_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gs_blk_kernel<NW>, LDU_WAVE * (NW + 1), lds));
    *perCU = n;
    return 0;
}

static inline int bk_width(const ldu_addr* a, int c)
{
    const BlockPlan* P = a->blocks;
    return a->losortStart[c + 1] - a->losortStart[c] + a->ownerStart[c + 1] - a->ownerStart[c]
           + (P && P->iface ? P->ifStart[c + 1] - P->ifStart[c] : 0);
}

// The grouping of sweep L (rows by their time T_L in the row-level DAG of the sweeps), its row and entry tables and the
// group-level times Phi.  Built when a smoothing call first asks for L + 1 sweeps (a level that is smoothed twice never pays
// for four groupings); needs the grouping of sweep L - 1.
static int bk_build_layout(ldu_addr* a, int L)
{
    BlockPlan* P = a->blocks;
    const int nC = a->nCells, nF = a->nFaces, nB = P->nBlocks;
    const bool verbose = getenv("LDU_VERBOSE") != nullptr;
    const std::vector<int>& blk = P->blk;
    const std::vector<int>& slot = P->slot;
    const std::vector<int>& rowBase = P->rowBase;
    const std::vector<int>& nLocal = P->blkNLocal;
    const std::vector<int>& ghostBase = P->ghostBase;
    const std::vector<int>& ghostCell = P->ghostCell;
    const std::vector<char>& exported = P->exported;
    std::vector<int> gslot(nC, -1);
    std::vector<int> RTprev;
    RTprev.swap(P->RTlast);
    std::vector<int> RT(nC);
    {
        BlockPlan::Layout& Y = P->lay[L];
        // T_L of every row (cells in ascending label order: lower neighbours first)
        for (int c = 0; c < nC; c++)
        {
            int t = L ? RTprev[c] : 0;
            if (L) for (int f = a->ownerStart[c]; f < a->ownerStart[c + 1]; f++) t = std::max(t, RTprev[a->u[f]]);
            // (a mirror - the cell across a processor-patch face - has no time of its own here: its rank sweeps in step)
            if (L && P->iface) for (int e = P->ifStart[c]; e < P->ifStart[c + 1]; e++) if (P->ifNbr[e] < nC) t = std::max(t, RTprev[P->ifNbr[e]]);
            for (int s = a->losortStart[c]; s < a->losortStart[c + 1]; s++) t = std::max(t, RT[a->l[a->losort[s]]]);
            RT[c] = t + 1;
        }
        const int maxRT = *std::max_element(RT.begin(), RT.end());
        // rows sorted by (block, T, lanes per row, level-ordered row): counting sort by T of the level-ordered rows, then by block
        std::vector<int> order(nC);
        {
            std::vector<long> st((size_t)maxRT + 2, 0);
            for (int c = 0; c < nC; c++) st[(size_t)RT[c] + 1]++;
            for (size_t i = 0; i + 1 < st.size(); i++) st[i + 1] += st[i];
            std::vector<int> byT(nC);
            for (int r = 0; r < nC; r++) { const int c = a->perm[r]; byT[(size_t)st[RT[c]]++] = c; }
            std::vector<int> pos(rowBase.begin(), rowBase.end() - 1);
            for (int i = 0; i < nC; i++) { const int c = byT[i]; order[pos[blk[c]]++] = c; }
        }
        auto lanesOf = [&](int c) { const int w = bk_width(a, c); return w <= 16 ? 1 : w <= 32 ? 2 : w <= 64 ? 4 : 8; };
        Y.grpOfCell.assign(nC, 0);
        long nLanes = 0, nEnt = 0;
        for (int b = 0; b < nB; b++)
        {
            int i = rowBase[b];
            while (i < rowBase[b + 1])
            {
                // rows of one T; inside it one group per lanes-per-row class (1, 2, 4, 8), at most 64 lanes each
                int iEnd = i;
                while (iEnd < rowBase[b + 1] && RT[order[iEnd]] == RT[order[i]]) iEnd++;
                // one group with the lanes-per-row of its widest row when that fits a wavefront (half as many tasks on the
                // agglomerated levels, where a third of the rows have more than 16 entries); otherwise one group per class
                int Tmax = 1;
                for (int t = i; t < iEnd; t++) Tmax = std::max(Tmax, lanesOf(order[t]));
                const bool uniform = (iEnd - i) * Tmax <= LDU_WAVE;
                for (int Tc = uniform ? Tmax : 1; Tc <= 8; Tc *= 2)
                {
                    int cnt = 0;
                    auto flush = [&]() {
                        if (!cnt) return;
                        const int stride = std::max(8, (cnt + 7) & ~7);
                        Y.grpBlk.push_back(b); Y.grpLane0.push_back((int)nLanes); Y.grpCnt.push_back(cnt); Y.grpT.push_back(Tc);
                        Y.grpEnt.push_back((int)nEnt); Y.grpStride.push_back(stride);
                        nLanes += cnt;
                        nEnt += 16L * stride;
                        cnt = 0;
                    };
                    for (int t = i; t < iEnd; t++)
                    {
                        const int c = order[t];
                        if (!uniform && lanesOf(c) != Tc) continue;
                        if (cnt + Tc > LDU_WAVE) flush();
                        Y.grpOfCell[c] = (int)Y.grpBlk.size();
                        cnt += Tc;
                    }
                    flush();
                    if (uniform) break;
                }
                i = iEnd;
            }
            if (nEnt > 0x7ff00000L) return 0;     // (entry offsets are 32-bit in the task records)
        }
        Y.nGroups = (int)Y.grpBlk.size();
        Y.nLanes = nLanes;
        Y.nEntries = nEnt + 16 * LDU_WAVE;
        // tables
        std::vector<int4> meta((size_t)nLanes);
        std::vector<unsigned> col((size_t)(Y.nEntries / 2), 0);
        std::vector<int> srcFace((size_t)Y.nEntries, -1);
        std::vector<int> fill(Y.nGroups, 0);     // lanes placed so far per group
        // (block by block: a block's rows, groups and entries are its own - ranges of blocks on host threads for large levels,
        //  each with its own ghost-slot look-up; the tables of the 3.1 M-cell level took most of its plan's 0.6 s per sweep)
        auto fillBlocks = [&](long b0, long b1) {
            std::vector<int> gslotT;
            std::vector<int>& gslot_ = (b0 == 0 && b1 == nB) ? gslot : gslotT;
            if (&gslot_ == &gslotT) gslotT.assign(nC, -1);
            std::vector<int> hslot(P->iface ? (size_t)nC + a->nPatchFaces : 0, -1);     // hist pair (first LDS slot) of a cell (or mirror) in the current block
            for (int b = (int)b0; b < (int)b1; b++)
            {
                for (int g = ghostBase[b]; g < ghostBase[b + 1]; g++) gslot_[ghostCell[g]] = nLocal[b] + (g - ghostBase[b]);
                if (P->iface)
                {
                    const int h0 = nLocal[b] + ghostBase[b + 1] - ghostBase[b];
                    for (int h = P->histBase[b]; h < P->histBase[b + 1]; h++) hslot[P->histCell[h]] = h0 + 2 * (h - P->histBase[b]);
                }
            for (int t = rowBase[b]; t < rowBase[b + 1]; t++)
            {
                const int c = order[t];
                const int g = Y.grpOfCell[c], Tc = Y.grpT[g], stride = Y.grpStride[g];
                const int lane0 = fill[g];
                fill[g] += Tc;
                // entries of a row: interface faces (patch, face order), lower neighbours, upper neighbours
                const int ni = P->iface ? P->ifStart[c + 1] - P->ifStart[c] : 0;
                const int nl = a->losortStart[c + 1] - a->losortStart[c], nn = ni + nl + a->ownerStart[c + 1] - a->ownerStart[c];
                for (int tl = 0; tl < Tc; tl++)
                {
                    const int fi = std::min(16, std::max(0, ni - 16 * tl));                       // interface entries of this lane
                    const int lo = std::min(16, std::max(0, ni + nl - 16 * tl)) - fi;             // lower entries of this lane
                    const int en = std::min(16, std::max(0, nn - 16 * tl));
                    int4 M;
                    M.x = a->iperm[c] | (exported[c] ? (int)0x80000000 : 0) | (tl == Tc - 1 ? 0x40000000 : 0);
                    M.y = slot[c] | (lo << 16) | (en << 21) | (fi << 26);
                    M.z = ni ? hslot[c] : -1;
                    M.w = ni ? P->ifIdx[c] : 0;
                    meta[(size_t)Y.grpLane0[g] + lane0 + tl] = M;
                }
                int q = 0;
                auto put = [&](int n, int code) {
                    const int tl = q >> 4, qq = q & 15;
                    // a lane's entries 2 p, 2 p + 1 are one 16-byte pair (pair p of lane l at grpEnt / 2 + p * stride + l), its
                    // column slots 8 r ... 8 r + 7 one 16-byte quad of words (quad r at grpEnt / 8 + r * stride + l): a task's
                    // rows arrive with 2 + 8 load instructions instead of 8 + 16 - groups hold ~10 of 64 lanes, what a
                    // task's fill costs is the number of instructions, not the bytes
                    const size_t e = 2 * ((size_t)(Y.grpEnt[g] / 2) + (size_t)(qq >> 1) * stride + (size_t)(lane0 + tl)) + (size_t)(qq & 1);
                    const unsigned sl = code < -1 ? (unsigned)hslot[n] : (unsigned)(blk[n] == b ? slot[n] : gslot_[n]);
                    const size_t e2 = 4 * ((size_t)(Y.grpEnt[g] / 8) + (size_t)(qq >> 3) * stride + (size_t)(lane0 + tl)) + (size_t)((qq >> 1) & 3);
                    col[e2] |= (qq & 1) ? sl << 16 : sl;
                    srcFace[e] = code;
                    q++;
                };
                for (int e = P->iface ? P->ifStart[c] : 0; e < (P->iface ? P->ifStart[c + 1] : 0); e++) put(P->ifNbr[e], -2 - P->ifPf[e]);
                for (int s = a->losortStart[c]; s < a->losortStart[c + 1]; s++) { const int f = a->losort[s]; put(a->l[f], f << 1); }
                for (int f = a->ownerStart[c]; f < a->ownerStart[c + 1]; f++) put(a->u[f], (f << 1) | 1);
            }
            }
        };
        {
            const int nT = nC >= 200000 ? (int)std::min<long>(std::min(8u, std::max(1u, std::thread::hardware_concurrency())), nB) : 1;
            if (nT <= 1) fillBlocks(0, nB);
            else
            {
                // ranges of blocks with about the same number of rows
                std::vector<std::thread> th;
                int b0 = 0;
                for (int t = 0; t < nT; t++)
                {
                    int b1 = b0;
                    const long want = (long)nC * (t + 1) / nT;
                    while (b1 < nB && (t == nT - 1 || rowBase[b1 + 1] <= want)) b1++;
                    if (t == nT - 1) b1 = nB;
                    if (b1 > b0) th.emplace_back(fillBlocks, (long)b0, (long)b1);
                    b0 = b1;
                }
                for (auto& x : th) x.join();
            }
        }
        // group-level times Phi: groups in ascending T (a group holds one T): all inputs of lower T are final
        Y.Phi.assign(Y.nGroups, 0);
        {
            const BlockPlan::Layout* Yp = L ? &P->lay[L - 1] : nullptr;
            std::vector<long> st((size_t)maxRT + 2, 0);
            for (int c = 0; c < nC; c++) st[(size_t)RT[c] + 1]++;
            for (size_t i = 0; i + 1 < st.size(); i++) st[i + 1] += st[i];
            std::vector<int> byT(nC);
            { std::vector<long> pos(st.begin(), st.end() - 1); for (int c = 0; c < nC; c++) byT[(size_t)pos[RT[c]]++] = c; }
            for (int tv = 0; tv <= maxRT; tv++)
            {
                for (long i = st[tv]; i < st[tv + 1]; i++)
                {
                    const int c = byT[(size_t)i], g = Y.grpOfCell[c];
                    int ph = Y.Phi[g];
                    if (Yp)
                    {
                        ph = std::max(ph, Yp->Phi[Yp->grpOfCell[c]] + 1);
                        for (int f = a->ownerStart[c]; f < a->ownerStart[c + 1]; f++) ph = std::max(ph, Yp->Phi[Yp->grpOfCell[a->u[f]]] + 1);
                        if (P->iface)
                            for (int e = P->ifStart[c]; e < P->ifStart[c + 1]; e++)
                            {
                                const int n = P->ifNbr[e];
                                if (n < nC) ph = std::max(ph, Yp->Phi[Yp->grpOfCell[n]] + 1);
                                else if ((int)P->remPhi[L - 1].size() == a->nPatchFaces) ph = std::max(ph, P->remPhi[L - 1][n - nC] + 1);
                            }
                    }
                    for (int s = a->losortStart[c]; s < a->losortStart[c + 1]; s++)
                        ph = std::max(ph, Y.Phi[Y.grpOfCell[a->l[a->losort[s]]]] + 1);
                    Y.Phi[g] = ph;
                }
            }
        }
        if (bk_upload(&Y.d_meta, meta, 64) || bk_upload(&Y.d_col, col) || bk_upload(&Y.d_srcFace, srcFace)) return -1;
        if (verbose)
            fprintf(stderr, "[ldugpu] block engine plan: sweep %d: row-level DAG %d steps; %d groups (%.1f lanes each), group-level DAG %d steps, "
                            "%.1f M entries (%.2f x the coefficients)\n", L, maxRT, Y.nGroups, (double)nLanes / std::max(1, Y.nGroups),
                    *std::max_element(Y.Phi.begin(), Y.Phi.end()) + 1, Y.nEntries / 1e6, (double)Y.nEntries / std::max(1, 2 * nF));
        P->RTlast.swap(RT);
    }
    P->nBuilt = L + 1;
    return 0;
}

static int bk_build(ldu_addr* a)
{
    if (a->blocks) return 0;
    ldu_ctx* ctx = a->ctx;
    BlockPlan* P = new BlockPlan();
    a->blocks = P;
    P->built = true;
    const int nC = a->nCells, nF = a->nFaces;
    // (levels with coupled patches have no one-workgroup engine to fall back on: from 64 cells)
    if (!ctx->blkEngine || nC < (a->nPatchFaces ? std::min(64, ctx->blkMinCells) : ctx->blkMinCells) || nC > ctx->blkMaxCells || nF == 0) return 0;
    const auto tB0 = std::chrono::steady_clock::now();
    const bool verbose = getenv("LDU_VERBOSE") != nullptr;
    if (a->nPatchFaces)
    {
        // coupled patches: cyclic ones only (the neighbour is a cell of this addressing); a processor patch's neighbour lives on
        // another rank - those levels stay on the level engines
        { const char* e = getenv("LDU_BLK_IFACE"); if (e && !atoi(e)) return 0; }      // LDU_BLK_IFACE=0: patched levels on the level engines
        for (const Patch& q : a->patches) if (q.nbrPatch < 0 && q.n) P->remote = true;
        // processor patches: through the peer windows (see "Remote interfaces"), where this addressing and the ones across its
        // patches have their regions there and every rank has said so (k_blocks_peer_decide)
        if (P->remote && !(a->peer && a->peer->bAll && a->peer->d_bdst && comm_peer_carries_halo(ctx))) return 0;
        P->pfCell.assign(a->nPatchFaces, 0);
        for (const Patch& q : a->patches) for (int i = 0; i < q.n; i++) P->pfCell[q.offset + i] = q.faceCells[i];
        P->ifStart.assign(nC + 1, 0);
        for (const Patch& q : a->patches) for (int c : q.faceCells) P->ifStart[c + 1]++;
        for (int c = 0; c < nC; c++) P->ifStart[c + 1] += P->ifStart[c];
        P->ifPf.resize(a->nPatchFaces); P->ifNbr.resize(a->nPatchFaces);
        {
            std::vector<int> pos(P->ifStart.begin(), P->ifStart.end() - 1);
            for (const Patch& q : a->patches)          // patch order, face order: the order of updateMatrixInterfaces
            {
                if (q.nbrPatch >= 0 && a->patches[q.nbrPatch].n != q.n) return 0;
                for (int i = 0; i < q.n; i++)
                {
                    const int c = q.faceCells[i];
                    P->ifPf[pos[c]] = q.offset + i;
                    // (the cell across a processor-patch face: its mirror, label nCells + patch face)
                    P->ifNbr[pos[c]++] = q.nbrPatch >= 0 ? a->patches[q.nbrPatch].faceCells[i] : nC + q.offset + i;
                }
            }
        }
        P->ifIdx.assign(nC, -1);
        for (int c = 0; c < nC; c++) if (P->ifStart[c + 1] > P->ifStart[c]) P->ifIdx[c] = P->nIf++;
        P->iface = true;
    }
    for (int c = 0; c < nC; c++)
        if (bk_width(a, c) > 128) return 0;       // (8 lanes x 16 entries per row)
    // How the matrix is cut.  Every block must be resident (see "Progress"), so a configuration is (compute wavefronts per
    // block, workgroups per CU): 7 + 1 wavefronts = 512 threads, one workgroup per CU, up to 160 KB of LDS = ~18 000 slots per
    // block, 256 blocks; or 3 + 1 wavefronts = 256 threads, two per CU, 80 KB each, 512 blocks.  Measured on the GAMG levels of
    // the 12.7 M-cell motorBike mesh (profiles/r05_block_engine_probe.log, ms per four sweeps with 7 / 3 wavefronts): 769 k
    // cells 1.57 / 1.77, 189 k equal, 46 k 0.85 / 0.75, 11 k 0.65 / 0.57 - seven from blkWideFrom cells, three below; the other
    // configuration is tried when the preferred one does not fit.  Blocks: nearly equal breadth-first blobs (partition_blobs).
    P->nLayouts = std::min(BK_NLAY, std::max(1, ctx->blkLayouts));
    std::vector<int> blk(nC);
    int nB = 0, nw = 7;
    std::vector<int> nLocal, slot(nC), rowBase;
    std::vector<int>& ghostBase = P->ghostBase;
    std::vector<int>& ghostCell = P->ghostCell;
    std::vector<unsigned char>& ghostLower = P->ghostLower;
    int maxSlots = 0, perCU = 0;
    const int pref = ctx->blkWaves == 3 || ctx->blkWaves == 7 ? ctx->blkWaves : (nC >= ctx->blkWideFrom ? 7 : 3);
    bool found = false;
    // Candidates 2 and 3: blobs of equal FOOTPRINT (cells + ghosts, partition_blobs_slots) at 92 % / 85 % of one workgroup's LDS,
    // 7 + 1 wavefronts, one per CU - for the levels whose equal-size blobs do not fit (3.1 M-cell level of the motorBike mesh:
    // 242 blobs of 12.9 k cells, the largest with 23.7 k slots = 1.42 x the mean, 18.2 k fit).  Levels above blkEqualMax cells
    // start there: the first two are known not to fit and cost a partition each.
    for (int cand = nC > ctx->blkEqualMax ? 2 : 0; cand < 4 && !found; cand++)
    {
        const bool bySlots = cand >= 2;
        if (bySlots && (P->iface || !ctx->blkBySlots)) break;
        nw = bySlots ? 7 : (cand == 0 ? pref : (pref == 7 ? 3 : 7));
        if (cand == 1 && (ctx->blkWaves == 3 || ctx->blkWaves == 7)) continue;      // forced
        int perCUregs = 0;
        if (nw == 7 ? bk_occupancy<7>(1024, &perCUregs) : bk_occupancy<3>(1024, &perCUregs)) return -1;
        if (perCUregs > ctx->blkMaxPerCU) perCUregs = ctx->blkMaxPerCU;
        if (perCUregs < 1) continue;
        const long cap = std::max<long>(1, (long)ctx->numCUs * perCUregs / std::max(1, ctx->deviceSharers));
        // as many blocks as fit at once (a few spare: the partitioner can come back with fewer), but not smaller than cmin cells
        const int cmin = nw == 3 ? std::min(ctx->blkCellsMin, 256) : ctx->blkCellsMin;
        long nParts = ctx->blkCells > 0 ? ((long)nC + ctx->blkCells - 1) / ctx->blkCells : std::min<long>((long)(0.97 * (double)cap), std::max<long>(1, nC / cmin));
        if (nParts < 1) nParts = 1;
        if (nParts > cap) { if (ctx->blkCells > 0) continue; nParts = cap; }
        if (bySlots)
        {
            const long slotT = (long)((cand == 2 ? 0.92 : 0.85) * (double)((BK_MAX_LDS - 80) / 9));
            nParts = cap;
            nB = partition_blobs_slots(nC, nF, a->l.data(), a->u.data(), slotT, (int)cap, blk.data());
            if (nB == -2) { if (verbose) fprintf(stderr, "[ldugpu] block engine: %d cells: more than %ld blobs of %ld slots\n", nC, cap, slotT); break; }
        }
        else
            nB = partition_blobs(nC, nF, a->l.data(), a->u.data(), (int)nParts, blk.data());
        if (nB < 1) return -1;
        // LDS slots of a block's rows: the order of the level-ordered numbering restricted to the block
        nLocal.assign(nB, 0);
        for (int r = 0; r < nC; r++) { const int c = a->perm[r]; slot[c] = nLocal[blk[c]]++; }
        // ghosts
        ghostBase.assign(nB + 1, 0);
        ghostCell.clear();
        ghostLower.clear();
        {
            std::vector<int> cellsOf(nC), start(nB + 1, 0);
            for (int c = 0; c < nC; c++) start[blk[c] + 1]++;
            for (int b = 0; b < nB; b++) start[b + 1] += start[b];
            { std::vector<int> pos(start.begin(), start.end() - 1); for (int c = 0; c < nC; c++) cellsOf[pos[blk[c]]++] = c; }
            std::vector<int> mark(nC, -1), gidx(nC, 0);
            for (int b = 0; b < nB; b++)
            {
                ghostBase[b] = (int)ghostCell.size();
                for (int t = start[b]; t < start[b + 1]; t++)
                {
                    const int c = cellsOf[t];
                    for (int s = a->losortStart[c]; s < a->losortStart[c + 1]; s++)
                    {
                        const int n = a->l[a->losort[s]];
                        if (blk[n] == b) continue;
                        if (mark[n] != b) { mark[n] = b; gidx[n] = (int)ghostCell.size(); ghostCell.push_back(n); ghostLower.push_back(0); }
                        ghostLower[gidx[n]] = 1;
                    }
                    for (int f = a->ownerStart[c]; f < a->ownerStart[c + 1]; f++)
                    {
                        const int n = a->u[f];
                        if (blk[n] == b) continue;
                        if (mark[n] != b) { mark[n] = b; gidx[n] = (int)ghostCell.size(); ghostCell.push_back(n); ghostLower.push_back(0); }
                    }
                }
            }
            ghostBase[nB] = (int)ghostCell.size();
        }
        // hist pairs (interfaces): a block's own interface cells (slot order), then the cells across its interfaces
        P->histBase.assign(nB + 1, 0);
        P->histCell.clear();
        if (P->iface)
        {
            std::vector<int> cellsOf(nC), start(nB + 1, 0);
            for (int c = 0; c < nC; c++) start[blk[c] + 1]++;
            for (int b = 0; b < nB; b++) start[b + 1] += start[b];
            { std::vector<int> pos(start.begin(), start.end() - 1); for (int r = 0; r < nC; r++) { const int c = a->perm[r]; cellsOf[pos[blk[c]]++] = c; } }
            std::vector<int> mark((size_t)nC + a->nPatchFaces, -1);
            for (int b = 0; b < nB; b++)
            {
                P->histBase[b] = (int)P->histCell.size();
                for (int t = start[b]; t < start[b + 1]; t++)
                    if (P->ifIdx[cellsOf[t]] >= 0) { mark[cellsOf[t]] = b; P->histCell.push_back(cellsOf[t]); }
                for (int t = start[b]; t < start[b + 1]; t++)
                {
                    const int c = cellsOf[t];
                    for (int e = P->ifStart[c]; e < P->ifStart[c + 1]; e++)
                    {
                        const int n = P->ifNbr[e];
                        if (mark[n] != b) { mark[n] = b; P->histCell.push_back(n); }
                    }
                }
            }
            P->histBase[nB] = (int)P->histCell.size();
        }
        maxSlots = 0;
        for (int b = 0; b < nB; b++)
            maxSlots = std::max(maxSlots, nLocal[b] + ghostBase[b + 1] - ghostBase[b] + 2 * (P->histBase[b + 1] - P->histBase[b]));
        const size_t lds = ((size_t)9 * maxSlots + 64 + 15) & ~(size_t)15;
        perCU = 0;
        if (lds <= BK_MAX_LDS && maxSlots < 65536)
        {
            if (nw == 7 ? bk_occupancy<7>(lds, &perCU) : bk_occupancy<3>(lds, &perCU)) return -1;
            if (perCU > ctx->blkMaxPerCU) perCU = ctx->blkMaxPerCU;
        }
        P->ldsBytes = lds;
        // (a GPU shared with other ranks' contexts: only this context's share of the workgroup slots may be counted on -
        //  the others' spinning workgroups hold theirs)
        found = perCU >= 1 && (long)nB <= (long)perCU * ctx->numCUs / std::max(1, ctx->deviceSharers);
        if (verbose)
            fprintf(stderr, "[ldugpu] block engine: %d cells, %d + 1 wavefronts per block: %d blocks (%ld asked for), largest %d slots (%zu B "
                            "of LDS), %d ghosts in all, %d workgroups per CU -> %ld resident%s\n", nC, nw, nB, nParts, maxSlots, lds,
                    (int)ghostCell.size(), perCU, (long)perCU * ctx->numCUs, found ? "" : ": does not fit");
    }
    if (!found) return 0;
    P->nw = nw;
    if (perCU < 1) return 0;
    P->capacity = (long)perCU * ctx->numCUs / std::max(1, ctx->deviceSharers);
    P->nBlocks = nB;
    P->maxSlots = maxSlots;
    P->blkNLocal = nLocal;
    P->nGhostTotal = (int)ghostCell.size();
    rowBase.assign(nB + 1, 0);
    for (int b = 0; b < nB; b++) rowBase[b + 1] = rowBase[b] + nLocal[b];
    std::vector<int> localRow(nC);
    for (int c = 0; c < nC; c++) localRow[rowBase[blk[c]] + slot[c]] = a->iperm[c];
    P->ghostRowH.resize(ghostCell.size());
    for (size_t g = 0; g < ghostCell.size(); g++) P->ghostRowH[g] = a->iperm[ghostCell[g]];
    // ghost slot of (block, cell): filled block by block below
    std::vector<char> exported(nC, 0);
    for (int f = 0; f < nF; f++)
        if (blk[a->l[f]] != blk[a->u[f]]) exported[a->l[f]] = exported[a->u[f]] = 1;

    P->blk.swap(blk);
    P->slot.swap(slot);
    P->rowBase = rowBase;
    P->exported.swap(exported);
    P->nBuilt = 0;
    // ---- upload
    std::vector<int4> blkInfo(nB);
    for (int b = 0; b < nB; b++) blkInfo[b] = make_int4(rowBase[b], nLocal[b], ghostBase[b], ghostBase[b + 1] - ghostBase[b]);
    if (bk_upload(&P->d_blk, blkInfo) || bk_upload(&P->d_localRow, localRow) || bk_upload(&P->d_ghostRow, P->ghostRowH)) return -1;
    if (P->iface)
    {
        std::vector<int2> b2(nB);
        for (int b = 0; b < nB; b++) b2[b] = make_int2(P->histBase[b], P->histBase[b + 1] - P->histBase[b]);
        std::vector<int> hr(P->histCell.size());
        for (size_t i = 0; i < hr.size(); i++) hr[i] = P->histCell[i] < nC ? a->iperm[P->histCell[i]] : -1 - (P->histCell[i] - nC);
        if (bk_upload(&P->d_blk2, b2) || bk_upload(&P->d_histRow, hr)) return -1;
        if (P->remote)
        {
            // the processor-patch faces of every interface cell
            std::vector<int> rs(P->nIf + 1, 0), rp;
            for (int c = 0; c < nC; c++)
                if (P->ifIdx[c] >= 0)
                {
                    for (int e = P->ifStart[c]; e < P->ifStart[c + 1]; e++)
                        if (P->ifNbr[e] >= nC) rp.push_back(P->ifPf[e]);
                    rs[P->ifIdx[c] + 1] = (int)rp.size();
                }
            // (ifIdx ascends with the cell label: rs is complete)
            if (bk_upload(&P->d_remStart, rs) || bk_upload(&P->d_remPf, rp)) return -1;
        }
    }
    // granules: one per row, then two per interface cell (by sweep parity)
    LDU_CHECK_HIP(hipMalloc((void**)&P->d_granule, sizeof(uint4) * (size_t)(nC + 1 + 2 * (size_t)P->nIf)));
    LDU_CHECK_HIP(ldu_memset_sync(P->d_granule, 0, sizeof(uint4) * (size_t)(nC + 1 + 2 * (size_t)P->nIf)));
    LDU_CHECK_HIP(hipMalloc((void**)&P->d_ticket, sizeof(unsigned) * 64));
    LDU_CHECK_HIP(ldu_memset_sync(P->d_ticket, 0, sizeof(unsigned) * 64));
    LDU_CHECK_HIP(hipMalloc((void**)&P->d_out, sizeof(double) * (size_t)(nC + 1)));
    P->gen = ctx->p2pGen;
    P->eligible = true;
    if (verbose)
    {
        long cut = 0;
        for (int f = 0; f < nF; f++) cut += P->blk[a->l[f]] != P->blk[a->u[f]];
        fprintf(stderr, "[ldugpu] block engine plan: %d cells in %d blocks (%ld resident; %d wavefronts + importer each, %zu B of LDS), "
                        "%.1f %% of the faces cut, %d ghosts, %d dependency levels; %.3f s\n",
                nC, nB, P->capacity, nw, P->ldsBytes, 100.0 * cut / std::max(1, nF), P->nGhostTotal, a->nLevels,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tB0).count());
    }
    return 0;
}

// the task and import lists of k sweeps
static int bk_tasks(ldu_addr* a, int k, const BlockPlan::Tasks** out)
{
    BlockPlan* P = a->blocks;
    auto it = P->tasks.find(k);
    if (it == P->tasks.end())
    {
        const int nB = P->nBlocks;
        while (P->nBuilt < std::min(k, P->nLayouts))
            if (bk_build_layout(a, P->nBuilt)) return -1;
        auto layOf = [&](int j) -> const BlockPlan::Layout& { return P->lay[std::min(j, P->nLayouts - 1)]; };
        // (with fewer layouts than sweeps the later sweeps reuse the last grouping: their Phi = the last layout's + a shift that
        //  keeps every dependency ascending - computed here by running the group recurrence once more per extra sweep)
        std::vector<std::vector<int>> Phi(k);
        for (int j = 0; j < k; j++)
        {
            if (j < P->nLayouts) { Phi[j] = P->lay[j].Phi; continue; }
            const BlockPlan::Layout& Y = layOf(j);
            // the same grouping as sweep j - 1: Phi_j(g) = 1 + max(Phi_j(lower groups), Phi_j-1(upper groups), Phi_j-1(g)); groups
            // of one T are independent, and the group ids ascend with (block, T) only - so iterate to the fixed point in T order
            // via the cells in ascending T order of that layout: its Phi is itself a valid ascending key
            std::vector<int> ord(Y.nGroups);
            for (int g = 0; g < Y.nGroups; g++) ord[g] = g;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return Y.Phi[x] < Y.Phi[y]; });
            std::vector<int> rank(Y.nGroups);
            for (int i = 0; i < Y.nGroups; i++) rank[ord[i]] = i;
            Phi[j].assign(Y.nGroups, 0);
            // cells grouped per group
            std::vector<int> gs(Y.nGroups + 1, 0), gc(a->nCells);
            for (int c = 0; c < a->nCells; c++) gs[Y.grpOfCell[c] + 1]++;
            for (int g = 0; g < Y.nGroups; g++) gs[g + 1] += gs[g];
            { std::vector<int> pos(gs.begin(), gs.end() - 1); for (int c = 0; c < a->nCells; c++) gc[pos[Y.grpOfCell[c]]++] = c; }
            for (int i = 0; i < Y.nGroups; i++)
            {
                const int g = ord[i];
                int ph = Phi[j - 1][g] + 1;
                for (int t = gs[g]; t < gs[g + 1]; t++)
                {
                    const int c = gc[t];
                    for (int f = a->ownerStart[c]; f < a->ownerStart[c + 1]; f++) ph = std::max(ph, Phi[j - 1][Y.grpOfCell[a->u[f]]] + 1);
                    for (int s = a->losortStart[c]; s < a->losortStart[c + 1]; s++) ph = std::max(ph, Phi[j][Y.grpOfCell[a->l[a->losort[s]]]] + 1);
                }
                Phi[j][g] = ph;
            }
        }
        int maxT = 0;
        long nTasks = 0;
        for (int j = 0; j < k; j++) { maxT = std::max(maxT, *std::max_element(Phi[j].begin(), Phi[j].end())); nTasks += layOf(j).nGroups; }
        // tasks: counting sort by Phi (inside one Phi the sweeps ascend), then dealt to the blocks in that order
        std::vector<long> start((size_t)maxT + 2, 0);
        for (int j = 0; j < k; j++) for (int v : Phi[j]) start[(size_t)v + 1]++;
        for (size_t i = 0; i + 1 < start.size(); i++) start[i + 1] += start[i];
        std::vector<std::pair<int, int>> order((size_t)nTasks);
        for (int j = 0; j < k; j++)
            for (int g = 0; g < layOf(j).nGroups; g++) order[(size_t)start[Phi[j][g]]++] = std::make_pair(j, g);
        // wavefront of a task: sweep j runs on wavefronts j * wps ... j * wps + wps - 1 of its block (round-robin inside the sweep)
        const int NW = P->nw;
        // (blkWavesPerSweep = 0: all tasks of a block round-robin over its wavefronts, whatever their sweep - measured best: a
        //  wavefront spends ~1.5 us per task even when nothing has to be waited for, and the other wavefronts hide that)
        int wps = a->ctx->blkWavesPerSweep;
        if (wps * k > NW) wps = std::max(1, NW / k);
        std::vector<int> seq((size_t)nB * k, 0), seqB(nB, 0);
        std::vector<int> waveOf((size_t)nTasks);
        std::vector<int> taskStart((size_t)nB * NW + 1, 0);
        for (size_t i = 0; i < order.size(); i++)
        {
            const int j = order[i].first, b = layOf(j).grpBlk[order[i].second];
            const int w = wps > 0 ? (j * wps + (seq[(size_t)b * k + j]++ % wps)) % NW : seqB[b]++ % NW;
            waveOf[i] = w;
            taskStart[(size_t)b * NW + w + 1]++;
        }
        for (size_t i = 0; i + 1 < taskStart.size(); i++) taskStart[i + 1] += taskStart[i];
        std::vector<int4> tasks((size_t)nTasks);
        {
            std::vector<int> pos(taskStart.begin(), taskStart.end() - 1);
            for (size_t i = 0; i < order.size(); i++)
            {
                const BlockPlan::Layout& Y = layOf(order[i].first);
                const int g = order[i].second;
                tasks[(size_t)pos[(size_t)Y.grpBlk[g] * NW + waveOf[i]]++] =
                    make_int4(Y.grpLane0[g], Y.grpCnt[g] | (Y.grpT[g] << 8) | (Y.grpStride[g] << 16), Y.grpEnt[g], order[i].first);
            }
        }
        // imports: a ghost that is a lower neighbour of a local row is needed with stamps 1 ... k, one that is only an
        // upper neighbour with stamps 1 ... k - 1 (sweep 0 reads the initial value); due when the producing task is done
        std::vector<int> impStart(nB + 1, 0);
        std::vector<int4> imps;
        {
            std::vector<std::pair<long, int4>> tmp;
            for (int b = 0; b < nB; b++)
            {
                tmp.clear();
                for (int g = P->ghostBase[b]; g < P->ghostBase[b + 1]; g++)
                {
                    const int top = P->ghostLower[g] ? k : k - 1;
                    for (int s = 1; s <= top; s++)
                        tmp.emplace_back(((long)Phi[s - 1][layOf(s - 1).grpOfCell[P->ghostCell[g]]] << 3) | s,
                                         make_int4(P->blkNLocal[b] + (g - P->ghostBase[b]), P->ghostRowH[g], s, 0));
                }
                if (P->iface)
                {
                    // the cells across this block's interfaces (hist pairs behind its own interface cells): the value of sweep
                    // s - 1 (stamp s) for the rows of sweep s, s = 1 ... k - 1, into the slot of parity s & 1
                    const int h0 = P->blkNLocal[b] + P->ghostBase[b + 1] - P->ghostBase[b];
                    for (int h = P->histBase[b]; h < P->histBase[b + 1]; h++)
                    {
                        const int n = P->histCell[h];
                        if (n >= a->nCells)
                        {
                            // a mirror: the rank across patch face n - nCells sends its initial value (stamp 0) and the one after
                            // each sweep but the last; due about when the local cell of the face has done the same sweep
                            const int pf = n - a->nCells, cl = P->pfCell[pf];
                            for (int s = 0; s <= k - 1; s++)
                            {
                                const bool have = s && (int)P->remPhi[std::min(s - 1, P->nLayouts - 1)].size() == a->nPatchFaces;
                                const long due = !s ? 0L : (have ? (long)P->remPhi[std::min(s - 1, P->nLayouts - 1)][pf]
                                                                 : (long)Phi[s - 1][layOf(s - 1).grpOfCell[cl]]);
                                tmp.emplace_back(s ? ((due << 3) | s) : 0L, make_int4(h0 + 2 * (h - P->histBase[b]) + (s & 1), -1 - pf, s, 0));
                            }
                            continue;
                        }
                        if (P->blk[n] == b) continue;          // (its own cells: written by the producing wavefront itself)
                        for (int s = 1; s <= k - 1; s++)
                            tmp.emplace_back(((long)Phi[s - 1][layOf(s - 1).grpOfCell[n]] << 3) | s,
                                             make_int4(h0 + 2 * (h - P->histBase[b]) + (s & 1), a->nCells + 1 + 2 * P->ifIdx[n] + (s & 1), s, 0));
                    }
                }
                std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<long, int4>& x, const std::pair<long, int4>& y) { return x.first < y.first; });
                for (auto& t : tmp) imps.push_back(t.second);
                impStart[b + 1] = (int)imps.size();
            }
        }
        BlockPlan::Tasks W;
        W.nTasks = nTasks;
        // the order in which workgroups take the blocks (ascending first Phi) and the most blocks that are open at one Phi:
        // what the chip must hold at once (see "Progress" at the top of the file)
        {
            std::vector<int> lo(nB, 0x7fffffff), hi(nB, -1);
            for (int j = 0; j < k; j++)
            {
                const BlockPlan::Layout& Y = layOf(j);
                for (int g = 0; g < Y.nGroups; g++)
                {
                    const int b = Y.grpBlk[g], ph = Phi[j][g];
                    lo[b] = std::min(lo[b], ph);
                    hi[b] = std::max(hi[b], ph);
                }
            }
            std::vector<int> ord(nB);
            for (int b = 0; b < nB; b++) ord[b] = b;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return lo[x] < lo[y]; });
            std::vector<int> delta((size_t)maxT + 3, 0);
            for (int b = 0; b < nB; b++) if (hi[b] >= 0) { delta[lo[b]]++; delta[hi[b] + 1]--; }
            int open = 0;
            for (int t = 0; t <= maxT + 1; t++) { open += delta[t]; W.openMax = std::max(W.openMax, open); }
            // (a margin: a workgroup that has finished its block holds its slot until its last wavefront has left)
            W.usable = nB <= P->capacity || (double)W.openMax <= 0.8 * (double)P->capacity;
            if (P->iface && k > P->nLayouts) W.usable = false;     // (every sweep needs its own tables: the parity of the hist slots)
            if (bk_upload(&W.d_order, ord)) return -1;
        }
        if (bk_upload(&W.d_tasks, tasks, 1) || bk_upload(&W.d_taskStart, taskStart) || bk_upload(&W.d_imps, imps, 1) ||
            bk_upload(&W.d_impStart, impStart))
            return -1;
        if (getenv("LDU_VERBOSE"))
            fprintf(stderr, "[ldugpu] block engine: %d cells, k = %d: %ld tasks, %zu imports, %d steps in the group-level DAG; %d blocks, at "
                            "most %d open at a time, %ld resident%s\n", a->nCells, k, nTasks, imps.size(), maxT + 1, nB, W.openMax, P->capacity,
                    W.usable ? "" : " - NOT on this engine");
        it = P->tasks.emplace(k, W).first;
    }
    *out = &it->second;
    return 0;
}

static int bk_values(ldu_addr* a, const double* levelVal, const double* bou, hipStream_t s, const double* out[BK_NLAY])
{
    BlockPlan* P = a->blocks;
    auto org = a->valOrigin.find(levelVal);
    if (org == a->valOrigin.end()) return 1;
    BlockPlan::Conv& C = P->conv[levelVal];
    for (int L = 0; L < P->nBuilt; L++)
    {
        if (!C.d[L]) LDU_CHECK_HIP(hipMalloc((void**)&C.d[L], sizeof(double) * (size_t)P->lay[L].nEntries));
        if (C.stamp[L] != val_stamp(a, levelVal))
        {
            const int grid = (int)std::min<long>((P->lay[L].nEntries + 255) / 256, 8192);
            bk_fill_kernel<<<grid, 256, 0, s>>>(P->lay[L].nEntries, P->lay[L].d_srcFace, org->second.first, org->second.second, bou, C.d[L]);
            C.stamp[L] = val_stamp(a, levelVal);
        }
    }
    for (int L = 0; L < BK_NLAY; L++) out[L] = C.d[std::min(L, P->nBuilt - 1)];
    return 0;
}

static thread_local bool tl_blkDeciding = false;

bool k_blocks_active(ldu_addr* a)
{
    ldu_ctx* ctx = a->ctx;
    if (!ctx->blkEngine || !ctx->sweepP2P || a->nCells < (a->nPatchFaces ? std::min(64, ctx->blkMinCells) : ctx->blkMinCells)
        || a->nCells > ctx->blkMaxCells) return false;
    if (addr_bg_pending(a)) return false;      // (its plan is being built on a host thread: the level engines meanwhile)
    // remote interfaces: only what every rank has agreed on (k_blocks_peer_decide builds the plan itself)
    if (a->peer && a->peerBlk != 1 && !tl_blkDeciding) return false;
    if (!a->blocks && bk_build(a)) return false;
    return a->blocks && a->blocks->eligible;
}

// Remote interfaces, the collective part: an addressing with processor patches runs its pipelined sweeps on the block engine when
// EVERY rank's plan for it exists (its own regions in the peer windows, the ranks' across its patches, a block plan that fits
// for 1 ... 4 sweeps) - the ranks across an interface must speak the same protocol.  Every rank calls this for the same
// addressings in the same order (ldu_gamg.cpp: gamg_decide_peer_smoothers; smooth_gs under LDU_BLK_PEER_FORCE for stand-alone
// smoothing calls); an addressing without remote faces does not veto.
int k_blocks_peer_decide(ldu_addr* a)
{
    ldu_ctx* ctx = a->ctx;
    if (a->peerBlkEpoch == ctx->commEpoch) return 0;
    a->peerBlkEpoch = ctx->commEpoch;
    if (!ctx->comm) { a->peerBlk = 0; return 0; }
    int ok = 1;
    if (a->peer)
    {
        ok = 0;
        const bool off = getenv("LDU_BLK_PEER") && !atoi(getenv("LDU_BLK_PEER"));
        // (small levels whose sweeps AND exchanges already run in one launch - gs_wg_peer_kernel, decided collectively before
        //  this - stay there)
        if (!off && a->peerWg != 1 && ctx->blkEngine && ctx->sweepP2P && comm_peer_carries_halo(ctx) && a->peer->bAll)
        {
            addr_bg_wait(a);
            tl_blkDeciding = true;
            ok = k_blocks_active(a) && a->blocks->remote ? 1 : 0;
            tl_blkDeciding = false;
        }
    }
    // 1. every rank has a block plan for this level (or no remote faces on it)
    if (comm_allreduce_min_int(ctx, &ok)) return -1;
    // 2. the per-sweep layouts, one after the other on all ranks together: layout L needs the group times of layout L - 1 of
    //    the cells ACROSS the processor patches (BlockPlan::remPhi) - exchanged face by face like the agglomeration's restrict
    //    maps (comm_exchange_ints); a rank whose own build fails keeps exchanging (zeros) and vetoes at the end
    if (ok && a->peer)
    {
        BlockPlan* P = a->blocks;
        int mine = 1;
        tl_blkDeciding = true;
        for (int L = 0; L < P->nLayouts; L++)
        {
            if (L)
            {
                std::vector<std::vector<int>> send(a->patches.size()), recv;
                const BlockPlan::Layout& Y = P->lay[L - 1];
                for (size_t p = 0; p < a->patches.size(); p++)
                {
                    send[p].assign(a->patches[p].n, 0);
                    if (mine && P->nBuilt >= L) for (int i = 0; i < a->patches[p].n; i++) send[p][i] = Y.Phi[Y.grpOfCell[a->patches[p].faceCells[i]]];
                }
                if (comm_exchange_ints(ctx, a->patches, send, recv)) { tl_blkDeciding = false; return -1; }
                P->remPhi[L - 1].assign(a->nPatchFaces, 0);
                for (size_t p = 0; p < a->patches.size(); p++)
                    for (int i = 0; i < a->patches[p].n && i < (int)recv[p].size(); i++) P->remPhi[L - 1][a->patches[p].offset + i] = recv[p][i];
            }
            // (a decision taken again - the carriers of the context changed -: the layouts are there, every rank's alike)
            if (mine && P->nBuilt <= L && (P->nBuilt != L || bk_build_layout(a, L) || P->nBuilt != L + 1)) mine = 0;
        }
        for (int k = 1; mine && k <= 4; k++)
        {
            const BlockPlan::Tasks* W = nullptr;
            if (bk_tasks(a, k, &W) || !W->usable) mine = 0;
        }
        tl_blkDeciding = false;
        ok = mine;
    }
    if (comm_allreduce_min_int(ctx, &ok)) return -1;
    a->peerBlk = (ok && a->peer) ? 1 : 0;
    if (!a->peerBlk && a->blocks && a->blocks->remote) a->blocks->eligible = false;
    if (getenv("LDU_VERBOSE") && a->peer)
        fprintf(stderr, "[ldugpu] level of %d cells, %d coupled faces: pipelined sweeps with remote interfaces on the block engine: %s\n",
                a->nCells, a->nPatchFaces, a->peerBlk ? "yes" : "no");
    return 0;
}

// the layouts' copies of the coefficients of `val`, filled on stream s ahead of the first sweep that asks for them (the
// coefficient chain of a GAMG solve, ldu_gamg.cpp: ensure_hierarchy); the sweep finds their stamps current.  Only plans and
// layouts that exist already (nothing is built here).  1 = nothing to fill
int k_blocks_prefill(ldu_addr* a, const double* val, const double* bou, hipStream_t s)
{
    if (addr_bg_pending(a)) return 1;
    BlockPlan* P = a->blocks;
    if (!P || !P->eligible || P->nBuilt <= 0) return 1;
    if (P->iface && !bou) return 1;
    const double* out[BK_NLAY];
    return bk_values(a, val, bou, s, out);
}

int k_blocks_prebuild(ldu_addr* a, int k)
{
    if (!k_blocks_active(a)) return 1;
    const BlockPlan::Tasks* W = nullptr;
    if (bk_tasks(a, k, &W)) return -1;
    return W->usable ? 0 : 2;     // 2: k sweeps of this addressing stay on the level engines (too many blocks open at a time)
}

int k_blocks_set_trace(unsigned long long* buf)
{
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bk_trace), &buf, sizeof(buf)));
    return 0;
}

// introspection (tools): {blocks, compute wavefronts per block, LDS bytes, ghosts, layouts, tasks of k sweeps (0 = not built)}
int k_blocks_info(ldu_addr* a, int k, long out[8])
{
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!k_blocks_active(a)) return 1;
    BlockPlan& P = *a->blocks;
    out[0] = P.nBlocks; out[1] = P.nw; out[2] = (long)P.ldsBytes; out[3] = P.nGhostTotal; out[4] = P.nLayouts;
    const BlockPlan::Tasks* W = nullptr;
    if (k >= 1 && k <= 4 && !bk_tasks(a, k, &W)) out[5] = W->nTasks;
    return 0;
}

// k pipelined GaussSeidel sweeps (k = 1 ... 4); 1 = not taken
int k_sweep_gs_blocks_if(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val, const double* bou);
int k_sweep_gs_blocks(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val)
{
    if (a->nPatchFaces) return 1;      // (with coupled patches: k_sweep_gs_blocks_if, which needs the interface coefficients)
    return k_sweep_gs_blocks_if(a, k, psi, rhs, diag, val, nullptr);
}

// ... with coupled (cyclic) interfaces: rhs = the SOURCE (the interface terms are entries of the rows), bou = interfaceBouCoeffs
int k_sweep_gs_blocks_if(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val, const double* bou)
{
    ldu_ctx* ctx = a->ctx;
    if (k <= 0 || k > 4) return 1;
    if (!k_blocks_active(a)) return 1;
    BlockPlan& P = *a->blocks;
    hipStream_t s = ctx->stream;
    BkTab T;
    if (a->valOrigin.find(val) == a->valOrigin.end()) return 1;   // a value array that was not filled from face-ordered coefficients
    const BlockPlan::Tasks* W = nullptr;
    if (bk_tasks(a, k, &W)) return -1;
    if (!W->usable) return 1;      // (more blocks open at a time than the chip holds: the level engines)
    {
        const int rc = bk_values(a, val, bou, s, T.val);
        if (rc) return rc;
    }
    if (P.gen != ctx->p2pGen)
    {
        LDU_CHECK_HIP(hipMemsetAsync(P.d_ticket, 0, sizeof(unsigned) * 64, s));
        P.ticketBase = 0;
        P.gen = ctx->p2pGen;
    }
    if (P.epoch > 0xffffff00u)
    {
        LDU_CHECK_HIP(hipMemsetAsync(P.d_granule, 0, sizeof(uint4) * (size_t)(a->nCells + 1 + 2 * (size_t)P.nIf), s));
        P.epoch = 0;
    }
    const unsigned tagBase = P.epoch;
    P.epoch += (unsigned)k;
    if (P.iface && !bou) return 1;
    T.blk = P.d_blk; T.localRow = P.d_localRow; T.ghostRow = P.d_ghostRow;
    T.blk2 = P.iface ? P.d_blk2 : nullptr; T.histRow = P.d_histRow; T.ifBase = a->nCells + 1;
    for (int L = 0; L < BK_NLAY; L++) { const BlockPlan::Layout& Y = P.lay[std::min(L, P.nBuilt - 1)]; T.meta[L] = Y.d_meta; T.col[L] = Y.d_col; }
    T.nLayouts = P.nLayouts;
    T.tasks = W->d_tasks; T.taskStart = W->d_taskStart; T.imps = W->d_imps; T.impStart = W->d_impStart;
    T.bdst = nullptr; T.bsrc = nullptr; T.nPF = a->nPatchFaces; T.S0 = 0; T.kSweeps = k;
    T.remStart = P.d_remStart; T.remPf = P.d_remPf;
    if (P.remote)
    {
        // k exchanges across the processor patches inside this launch: tags bSeq + 1 ... bSeq + k, the same on every rank
        // (each launches the same smoothing calls on this level)
        if (!a->peer || !a->peer->d_bdst || a->peerBlk != 1) return 1;
        T.bdst = a->peer->d_bdst; T.bsrc = a->peer->d_bsrc;
        T.S0 = a->peer->bSeq + 1u;
        a->peer->bSeq += (unsigned)k;
        ctx->nHaloExchanges += k;
        ctx->nHaloOverlapped += k;
        bk_init_export_kernel<<<std::min((a->nPatchFaces + 255) / 256, 1024), 256, 0, s>>>(a->nPatchFaces, a->d_pfCell, psi, T.bdst, T.S0);
    }
    ctx->profStart(a, 4);
    if (P.nw == 7)
        gs_blk_kernel<7><<<P.nBlocks, LDU_WAVE * 8, P.ldsBytes, s>>>(T, P.nBlocks, W->d_order, P.d_granule, tagBase, P.d_ticket, P.ticketBase,
                                                                     ctx->d_abort, psi, P.d_out, rhs, diag);
    else
        gs_blk_kernel<3><<<P.nBlocks, LDU_WAVE * 4, P.ldsBytes, s>>>(T, P.nBlocks, W->d_order, P.d_granule, tagBase, P.d_ticket, P.ticketBase,
                                                                     ctx->d_abort, psi, P.d_out, rhs, diag);
    bk_copy_kernel<<<(int)std::min<long>(((long)a->nCells + 255) / 256, 4096), 256, 0, s>>>(a->nCells, P.d_out, psi);
    ctx->profStop(a, 4);
    P.ticketBase += (unsigned)P.nBlocks;
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}
