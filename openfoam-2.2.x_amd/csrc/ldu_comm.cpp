// Communication layer: one rank per GPU / sub-domain.
//
// Backend 1 (production): RCCL over xGMI.  Replaces the MPI calls reached from the hot path
// (SURVEY.md 2.4):
//   reduce(scalar, sumOp)            -> ncclAllReduce on device-resident scalars (stays on the stream)
//   processor-patch Isend/Irecv      -> grouped ncclSend/ncclRecv of the packed halo buffers
//   restrict-map exchange (setup)    -> ncclSend/ncclRecv of int32 labels
//   continueAgglomerating and-reduce -> ncclAllReduce(min) of one int
// All of it is enqueued on the compute stream, ordered with the kernels without host syncs.
//
// Backend 2 (tests on a single GPU): "local" group - N contexts of ONE process, each driven by
// its own host thread, exchange through host-side barriers and device-to-device copies.  Same
// call sites, same semantics, no RCCL; lets the multi-rank algorithm be checked against the
// oracle's serial emulation on a 1-GPU box.
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unistd.h>

#include "ldu_internal.hpp"

struct LocalGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int count = 0;
    long gen = 0;
    std::vector<ldu_ctx*> ctx;
    std::vector<const ldu_addr*> addr;
    std::vector<std::vector<double>> dbuf;
    std::vector<const std::vector<std::vector<int>>*> isend;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const long g = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

// Backend 3: "peer" - windows of fine-grained device memory mapped by every rank of the node (hipIpc between
// processes, the plain pointer inside one process); halo values and partial sums are STORED into the neighbour's
// window by the producing kernel (ldu_peer.hip).  Set-up-time exchanges (window handles, region offsets, restrict
// maps, the and-reduce of continueAgglomerating) travel through an out-of-band callback of the host application -
// OpenFOAM's own Pstream in the shim, torch.distributed in the Python tests - exactly where the reference sends them
// (processorGAMGInterface.C:137-154 uses the same Pstream as everything else).
struct PeerWindow {
    uint4* base = nullptr;
    size_t bytes = 0;
    std::vector<uint4*> peer;                          // [nRanks] every rank's window as mapped here
    std::vector<bool> opened;                          // mapped with hipIpcOpenMemHandle (to be closed)
    std::map<size_t, size_t> freeList;                 // offset -> bytes (first fit, coalescing)
    size_t redOff = 0;                                 // [2][LDU_MAX_PEERS][16] granules of the all-reduce
    unsigned redSeq = 0;
    size_t redOffK = 0;                                // the same for collectives done INSIDE a kernel (own sequence, on the device)
    unsigned* d_redSeqK = nullptr;
    ldu_oob_exchange_fn oob = nullptr;
    void* oobUser = nullptr;
    size_t alloc(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        for (auto it = freeList.begin(); it != freeList.end(); ++it)
            if (it->second >= bytes)
            {
                const size_t off = it->first, rest = it->second - bytes;
                freeList.erase(it);
                if (rest) freeList[off + bytes] = rest;
                return off;
            }
        return (size_t)-1;
    }
    void release(size_t off, size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        auto it = freeList.emplace(off, bytes).first;
        auto nx = std::next(it);
        if (nx != freeList.end() && it->first + it->second == nx->first) { it->second += nx->second; freeList.erase(nx); }
        if (it != freeList.begin())
        {
            auto pv = std::prev(it);
            if (pv->first + pv->second == it->first) { pv->second += it->second; freeList.erase(it); }
        }
    }
};

struct ldu_comm_impl {
    ncclComm_t comm = nullptr;
    LocalGroup* local = nullptr;
    PeerWindow* peer = nullptr;
    bool peerHalo = false, peerReduce = false;   // which operations the peer backend carries (LDU_HALO / LDU_REDUCE)
    int* d_ibuf = nullptr;       // staging for int exchanges (RCCL)
    size_t ibufCap = 0;
    char* d_gbuf = nullptr;      // staging for comm_allgather_host (RCCL)
    size_t gbufCap = 0;
};

static int paired_patch(const std::vector<Patch>& mine, int p, const std::vector<Patch>& theirs, int me);
static int peer_oob(ldu_ctx* ctx, const std::vector<int>& peers, const std::vector<const void*>& send,
                    const std::vector<int64_t>& sendBytes, const std::vector<void*>& recv, const std::vector<int64_t>& recvBytes);
static std::mutex g_groupsMu;
static std::map<int, LocalGroup*> g_groups;

#define LDU_CHECK_NCCL(expr)                                                       \
    do {                                                                           \
        ncclResult_t _r = (expr);                                                  \
        if (_r != ncclSuccess) {                                                   \
            ldu_set_error(std::string(#expr) + ": " + ncclGetErrorString(_r));     \
            return -1;                                                             \
        }                                                                          \
    } while (0)

extern "C" int ldu_comm_unique_id(uint8_t id[128])
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    LDU_CHECK_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, 128);
    return 0;
}

extern "C" int ldu_ctx_comm_init(ldu_ctx* ctx, int rank, int nRanks, const uint8_t id[128])
{
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, 128);
    if (!ctx->comm) ctx->comm = new ldu_comm_impl();
    if (ctx->comm->peer && (ctx->rank != rank || ctx->nRanks != nRanks))
    {
        ldu_set_error("ldu_ctx_comm_init: rank / size differ from the peer windows'");
        return -2;
    }
    if (ctx->comm->comm) { ldu_set_error("ldu_ctx_comm_init: the context already has an RCCL communicator"); return -2; }
    LDU_CHECK_NCCL(ncclCommInitRank(&ctx->comm->comm, nRanks, u, rank));
    ctx->rank = rank;
    ctx->nRanks = nRanks;
    if (ctx->comm->peer)
    {
        // peer windows first, communicator second: RCCL becomes the default carrier, as ldugpu.h promises for either order
        // (LDU_HALO=p2p / LDU_REDUCE=p2p keep the halo exchanges / the global sums on peer stores)
        const char* eh = getenv("LDU_HALO");
        const char* er = getenv("LDU_REDUCE");
        ctx->comm->peerHalo = eh && !strcmp(eh, "p2p");
        ctx->comm->peerReduce = er && !strcmp(er, "p2p");
        ctx->commEpoch++;
    }
    return 0;
}

extern "C" int ldu_ctx_comm_init_local(ldu_ctx* ctx, int rank, int nRanks, int groupId)
{
    std::lock_guard<std::mutex> lk(g_groupsMu);
    LocalGroup*& G = g_groups[groupId];
    if (!G)
    {
        G = new LocalGroup();
        G->n = nRanks;
        G->ctx.assign(nRanks, nullptr);
        G->addr.assign(nRanks, nullptr);
        G->dbuf.assign(nRanks, std::vector<double>(S_BANK, 0.0));
        G->isend.assign(nRanks, nullptr);
    }
    if (G->n != nRanks) { ldu_set_error("local group size mismatch"); return -1; }
    G->ctx[rank] = ctx;
    ctx->comm = new ldu_comm_impl();
    ctx->comm->local = G;
    ctx->rank = rank;
    ctx->nRanks = nRanks;
    return 0;
}


// ---------------------------------------------------------------- peer backend: bootstrap and per-addressing set-up

static int peer_oob(ldu_ctx* ctx, const std::vector<int>& peers, const std::vector<const void*>& send,
                    const std::vector<int64_t>& sendBytes, const std::vector<void*>& recv, const std::vector<int64_t>& recvBytes)
{
    PeerWindow* W = ctx->comm->peer;
    if (peers.empty()) return 0;
    if (!W->oob) { ldu_set_error("peer backend: no out-of-band exchange callback"); return -4; }
    const int rc = W->oob(W->oobUser, (int32_t)peers.size(), peers.data(), send.data(), sendBytes.data(), recv.data(),
                          recvBytes.data());
    if (rc) { ldu_set_error("peer backend: the out-of-band exchange callback failed (" + std::to_string(rc) + ")"); return -4; }
    return 0;
}

struct PeerHello { int64_t pid; uint64_t ptr; int32_t device; int32_t pad; hipIpcMemHandle_t handle; };

extern "C" int ldu_ctx_comm_init_peer(ldu_ctx* ctx, int rank, int nRanks, ldu_oob_exchange_fn oob, void* user)
{
    if (nRanks < 1 || nRanks > LDU_MAX_PEERS || rank < 0 || rank >= nRanks)
    {
        ldu_set_error("ldu_ctx_comm_init_peer: 1 ... 16 ranks");
        return -2;
    }
    if (nRanks > 1 && !oob) { ldu_set_error("ldu_ctx_comm_init_peer: an out-of-band exchange callback is required"); return -2; }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    if (!ctx->comm) ctx->comm = new ldu_comm_impl();
    if (ctx->comm->local) { ldu_set_error("ldu_ctx_comm_init_peer: the context already has a local group"); return -2; }
    if (ctx->comm->comm && (ctx->rank != rank || ctx->nRanks != nRanks))
    {
        ldu_set_error("ldu_ctx_comm_init_peer: rank / size differ from the RCCL communicator's");
        return -2;
    }
    if (ctx->comm->peer) { ldu_set_error("ldu_ctx_comm_init_peer: the context already has peer windows"); return -2; }
    PeerWindow* W = new PeerWindow();
    ctx->comm->peer = W;     // (comm_destroy releases whatever of it exists, also after an error return below)
    W->oob = oob;
    W->oobUser = user;
    size_t mb = 256;
    if (const char* e = getenv("LDU_PEER_WINDOW_MB")) mb = (size_t)std::max(8, atoi(e));
    W->bytes = mb << 20;
    // fine-grained: coherent across agents while kernels run (coarse-grained memory is only guaranteed at kernel
    // boundaries); what RCCL allocates for its own peer-to-peer buffers
    LDU_CHECK_HIP(hipExtMallocWithFlags((void**)&W->base, W->bytes, hipDeviceMallocFinegrained));
    LDU_CHECK_HIP(ldu_memset_sync(W->base, 0, W->bytes));
    LDU_CHECK_HIP(hipDeviceSynchronize());
    W->freeList[0] = W->bytes;
    W->redOff = W->alloc(sizeof(uint4) * 2 * LDU_MAX_PEERS * 16);
    W->redOffK = W->alloc(sizeof(uint4) * 2 * LDU_MAX_PEERS * 16);
    if (W->redOff == (size_t)-1 || W->redOffK == (size_t)-1)
    {
        ldu_set_error("ldu_ctx_comm_init_peer: the window is too small for the reduction regions (LDU_PEER_WINDOW_MB)");
        return -1;
    }
    LDU_CHECK_HIP(hipMalloc((void**)&W->d_redSeqK, sizeof(unsigned)));
    LDU_CHECK_HIP(ldu_memset_sync(W->d_redSeqK, 0, sizeof(unsigned)));
    W->peer.assign(nRanks, nullptr);
    W->opened.assign(nRanks, false);
    W->peer[rank] = W->base;
    ctx->rank = rank;
    ctx->nRanks = nRanks;
    if (nRanks > 1)
    {
        PeerHello me;
        memset(&me, 0, sizeof(me));
        me.pid = (int64_t)getpid();
        me.ptr = (uint64_t)(uintptr_t)W->base;
        me.device = ctx->device;
        {
            // which physical GPU (the ordinal means nothing across processes with their own HIP_VISIBLE_DEVICES)
            int dom = 0, bus = 0, dev = 0;
            (void)hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, ctx->device);
            (void)hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, ctx->device);
            (void)hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, ctx->device);
            me.pad = (int32_t)(((unsigned)dom << 16) ^ ((unsigned)bus << 8) ^ (unsigned)dev) | 0x40000000;
        }
        LDU_CHECK_HIP(hipIpcGetMemHandle(&me.handle, W->base));
        std::vector<PeerHello> all(nRanks);
        std::vector<int> peers;
        std::vector<const void*> sp;
        std::vector<void*> rp;
        std::vector<int64_t> nb;
        for (int r = 0; r < nRanks; r++)
            if (r != rank) { peers.push_back(r); sp.push_back(&me); rp.push_back(&all[r]); nb.push_back(sizeof(PeerHello)); }
        if (peer_oob(ctx, peers, sp, nb, rp, nb)) return -1;
        // ranks on THIS GPU: the engines whose progress argument counts resident workgroups (ldu_blocks.hip) may count on
        // their share of the chip only - another rank's spinning workgroups hold the rest
        int sharers = 1;
        for (int r = 0; r < nRanks; r++) if (r != rank && all[r].pad == me.pad) sharers++;
        if (sharers > ctx->deviceSharers) ctx->deviceSharers = sharers;
        for (int r = 0; r < nRanks; r++)
        {
            if (r == rank) continue;
            if (all[r].pid == me.pid) { W->peer[r] = (uint4*)(uintptr_t)all[r].ptr; continue; }   // same process: the pointer itself
            void* p = nullptr;
            LDU_CHECK_HIP(hipIpcOpenMemHandle(&p, all[r].handle, hipIpcMemLazyEnablePeerAccess));
            W->peer[r] = (uint4*)p;
            W->opened[r] = true;
        }
    }
    // what travels by peer stores: everything unless an RCCL communicator exists as well, in which case RCCL stays the
    // default and LDU_HALO=p2p / LDU_REDUCE=p2p move the halo exchanges / the global sums over
    const bool haveRccl = ctx->comm->comm != nullptr;
    const char* eh = getenv("LDU_HALO");
    const char* er = getenv("LDU_REDUCE");
    ctx->comm->peerHalo = !haveRccl || (eh && !strcmp(eh, "p2p"));
    ctx->comm->peerReduce = !haveRccl || (er && !strcmp(er, "p2p"));
    if (const char* e = getenv("LDU_PEER_TIMEOUT_S")) if (k_peer_set_timeout(atof(e))) return -1;
    return 0;
}

// out[0] ranks of the RCCL communicator (ncclCommCount; 0 = none), [1] ranks whose windows are mapped (0 = no peer
// backend), [2] / [3] halo exchanges / global sums travel by peer stores
extern "C" int ldu_ctx_comm_info(const ldu_ctx* ctx, int32_t out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!ctx->comm) return 0;
    if (ctx->comm->comm)
    {
        int n = 0;
        LDU_CHECK_NCCL(ncclCommCount(ctx->comm->comm, &n));
        out[0] = n;
    }
    if (ctx->comm->peer) out[1] = (int32_t)ctx->comm->peer->peer.size();
    out[2] = ctx->comm->peerHalo;
    out[3] = ctx->comm->peerReduce;
    return 0;
}

// which operations the peer backend carries (0 / 1 each); both need ldu_ctx_comm_init_peer first
extern "C" int ldu_ctx_comm_select(ldu_ctx* ctx, int peerHalo, int peerReduce)
{
    if (!ctx->comm || !ctx->comm->peer) { ldu_set_error("ldu_ctx_comm_select: no peer windows (ldu_ctx_comm_init_peer)"); return -2; }
    if ((!peerHalo || !peerReduce) && !ctx->comm->comm)
    {
        ldu_set_error("ldu_ctx_comm_select: no RCCL communicator to carry what the peer backend does not");
        return -2;
    }
    ctx->comm->peerHalo = peerHalo != 0;
    ctx->comm->peerReduce = peerReduce != 0;
    ctx->commEpoch++;
    return 0;
}

// Plan time, every addressing with processor patches: a receive region in my window ([patch][parity][n] granules), its
// offsets told to the neighbours out of band, theirs received, and the per-face pointer tables uploaded.
int comm_peer_setup_addr(ldu_addr* a)
{
    ldu_ctx* ctx = a->ctx;
    if (!ctx->comm || !ctx->comm->peer) return 0;
    bool remote = false;
    for (auto& P : a->patches) if (P.nbrPatch < 0 && P.n) remote = true;
    // (an addressing without remote faces still takes part in no exchange: its neighbours have none towards it either)
    if (!remote) return 0;
    PeerWindow* W = ctx->comm->peer;
    PeerHalo* H = new PeerHalo();
    a->peer = H;
    const int nPF = a->nPatchFaces;
    size_t granules = 0;
    std::vector<size_t> myOff(a->patches.size(), 0);
    for (size_t p = 0; p < a->patches.size(); p++)
        if (a->patches[p].nbrPatch < 0) { myOff[p] = granules; granules += 2 * (size_t)a->patches[p].n; }
    H->winBytes = granules * sizeof(uint4);
    H->winOff = W->alloc(H->winBytes);
    if (H->winOff == (size_t)-1)
    {
        ldu_set_error("peer backend: window exhausted (LDU_PEER_WINDOW_MB, default 256)");
        return -5;
    }
    // kernel-private regions: coarsest-level candidates (<= 64 cells) and the levels of the one-workgroup engine
    const bool small = a->nCells <= LDU_PEERK_MAXCELLS;
    if (small)
    {
        H->kWinBytes = H->winBytes;
        H->kWinOff = W->alloc(H->kWinBytes);
        if (H->kWinOff == (size_t)-1) { ldu_set_error("peer backend: window exhausted (LDU_PEER_WINDOW_MB, default 256)"); return -5; }
        LDU_CHECK_HIP(hipMemsetAsync((char*)W->base + H->kWinOff, 0, H->kWinBytes, ctx->stream));
        LDU_CHECK_HIP(hipMalloc((void**)&H->d_kseq, sizeof(unsigned)));
        LDU_CHECK_HIP(hipMemsetAsync(H->d_kseq, 0, sizeof(unsigned), ctx->stream));
    }
    // the block engine's own regions (pipelined sweeps with remote interfaces): addressings in its range of sizes
    const bool blkRange = ctx->blkEngine && a->nCells >= 64 && a->nCells <= ctx->blkMaxCells && !getenv("LDU_BLK_PEER_OFF");
    if (blkRange)
    {
        H->bWinBytes = H->winBytes;
        H->bWinOff = W->alloc(H->bWinBytes);
        if (H->bWinOff == (size_t)-1) { H->bWinBytes = 0; H->bWinOff = 0; }     // (window full: this addressing stays on the level engines)
        else LDU_CHECK_HIP(hipMemsetAsync((char*)W->base + H->bWinOff, 0, H->bWinBytes, ctx->stream));
    }
    const bool haveB = H->bWinBytes > 0;
    // a region that is reused must not hold tags of its previous owner that a new sequence could reach: zero it
    LDU_CHECK_HIP(hipMemsetAsync((char*)W->base + H->winOff, 0, H->winBytes, ctx->stream));
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    // tell each neighbour where its k-th patch towards me lands: one message per neighbour rank, patch order
    std::map<int, std::vector<int>> byRank;
    for (int p = 0; p < (int)a->patches.size(); p++)
        if (a->patches[p].nbrPatch < 0 && a->patches[p].n > 0) byRank[a->patches[p].nbrRank].push_back(p);
    std::vector<int> peers;
    std::vector<std::vector<int64_t>> sb, rb;
    for (auto& kv : byRank)
    {
        std::vector<int64_t> buf;
        for (int p : kv.second)
        {
            buf.push_back((int64_t)(H->winOff / sizeof(uint4) + myOff[p]));
            buf.push_back((int64_t)a->patches[p].n);
            buf.push_back(small ? (int64_t)(H->kWinOff / sizeof(uint4) + myOff[p]) : (int64_t)-1);
            buf.push_back(haveB ? (int64_t)(H->bWinOff / sizeof(uint4) + myOff[p]) : (int64_t)-1);
        }
        peers.push_back(kv.first);
        rb.emplace_back(buf.size());
        sb.push_back(std::move(buf));
    }
    {
        std::vector<const void*> sp;
        std::vector<void*> rp;
        std::vector<int64_t> nb;
        for (size_t i = 0; i < peers.size(); i++)
        {
            sp.push_back(sb[i].data()); rp.push_back(rb[i].data()); nb.push_back((int64_t)(sizeof(int64_t) * sb[i].size()));
        }
        if (ctx->nRanks > 1)
        {
            if (peer_oob(ctx, peers, sp, nb, rp, nb)) return -1;
        }
        else rb = sb;   // one rank whose patches face each other (projection / tests): handled below
    }
    std::vector<uint4*> dst(2 * (size_t)nPF, nullptr), kdst(2 * (size_t)nPF, nullptr), bdst(2 * (size_t)nPF, nullptr);
    std::vector<const uint4*> src(2 * (size_t)nPF, nullptr), ksrc(2 * (size_t)nPF, nullptr), bsrc(2 * (size_t)nPF, nullptr);
    H->kAll = small;
    H->bAll = haveB;
    size_t i = 0;
    for (auto& kv : byRank)
    {
        const int nbr = kv.first;
        for (size_t k = 0; k < kv.second.size(); k++)
        {
            const int p = kv.second[k];
            const Patch& P = a->patches[p];
            int64_t roff = rb[i][4 * k], rn = rb[i][4 * k + 1], rkoff = rb[i][4 * k + 2], rboff = rb[i][4 * k + 3];
            if (ctx->nRanks == 1)
            {
                // self-coupled: the k-th patch towards "rank 0" pairs with the patch paired_patch names (itself when alone)
                const int q = paired_patch(a->patches, p, a->patches, ctx->rank);
                const int qq = q < 0 ? p : q;
                roff = (int64_t)(H->winOff / sizeof(uint4) + myOff[qq]);
                rkoff = small ? (int64_t)(H->kWinOff / sizeof(uint4) + myOff[qq]) : -1;
                rboff = haveB ? (int64_t)(H->bWinOff / sizeof(uint4) + myOff[qq]) : -1;
                rn = a->patches[qq].n;
            }
            if (rkoff < 0) H->kAll = false;
            if (rboff < 0) H->bAll = false;
            if (rn != P.n)
            {
                ldu_set_error("peer backend: patch sizes differ between neighbours (" + std::to_string(P.n) + " vs " +
                              std::to_string(rn) + ")");
                return -4;
            }
            for (int par = 0; par < 2; par++)
                for (int f = 0; f < P.n; f++)
                {
                    dst[(size_t)par * nPF + P.offset + f] = W->peer[nbr] + roff + (size_t)par * P.n + f;
                    src[(size_t)par * nPF + P.offset + f] = W->base + H->winOff / sizeof(uint4) + myOff[p] + (size_t)par * P.n + f;
                    if (small && rkoff >= 0)
                    {
                        kdst[(size_t)par * nPF + P.offset + f] = W->peer[nbr] + rkoff + (size_t)par * P.n + f;
                        ksrc[(size_t)par * nPF + P.offset + f] = W->base + H->kWinOff / sizeof(uint4) + myOff[p] + (size_t)par * P.n + f;
                    }
                    if (haveB && rboff >= 0)
                    {
                        bdst[(size_t)par * nPF + P.offset + f] = W->peer[nbr] + rboff + (size_t)par * P.n + f;
                        bsrc[(size_t)par * nPF + P.offset + f] = W->base + H->bWinOff / sizeof(uint4) + myOff[p] + (size_t)par * P.n + f;
                    }
                }
        }
        i++;
    }
    LDU_CHECK_HIP(hipMalloc((void**)&H->d_dst, sizeof(uint4*) * dst.size()));
    LDU_CHECK_HIP(hipMalloc((void**)&H->d_src, sizeof(uint4*) * src.size()));
    LDU_CHECK_HIP(hipMemcpy(H->d_dst, dst.data(), sizeof(uint4*) * dst.size(), hipMemcpyHostToDevice));
    LDU_CHECK_HIP(hipMemcpy(H->d_src, src.data(), sizeof(uint4*) * src.size(), hipMemcpyHostToDevice));
    if (haveB)
    {
        LDU_CHECK_HIP(hipMalloc((void**)&H->d_bdst, sizeof(uint4*) * bdst.size()));
        LDU_CHECK_HIP(hipMalloc((void**)&H->d_bsrc, sizeof(uint4*) * bsrc.size()));
        LDU_CHECK_HIP(hipMemcpy(H->d_bdst, bdst.data(), sizeof(uint4*) * bdst.size(), hipMemcpyHostToDevice));
        LDU_CHECK_HIP(hipMemcpy(H->d_bsrc, bsrc.data(), sizeof(uint4*) * bsrc.size(), hipMemcpyHostToDevice));
    }
    if (small)
    {
        LDU_CHECK_HIP(hipMalloc((void**)&H->d_kdst, sizeof(uint4*) * kdst.size()));
        LDU_CHECK_HIP(hipMalloc((void**)&H->d_ksrc, sizeof(uint4*) * ksrc.size()));
        LDU_CHECK_HIP(hipMemcpy(H->d_kdst, kdst.data(), sizeof(uint4*) * kdst.size(), hipMemcpyHostToDevice));
        LDU_CHECK_HIP(hipMemcpy(H->d_ksrc, ksrc.data(), sizeof(uint4*) * ksrc.size(), hipMemcpyHostToDevice));
    }
    return 0;
}

// the peer windows of this context are mapped on every rank (whichever carrier the halo exchanges between launches use: the
// block engine's in-launch interface stores only need the windows)
bool comm_peer_carries_halo(const ldu_ctx* ctx)
{
    return ctx->comm && ctx->comm->peer;
}

bool comm_peer_kernel_comm(ldu_ctx* ctx, PeerKernelComm* out)
{
    if (!ctx->comm || !ctx->comm->peer || !ctx->comm->peerHalo || !ctx->comm->peerReduce) return false;
    PeerWindow* W = ctx->comm->peer;
    for (int r = 0; r < LDU_MAX_PEERS; r++) out->P.win[r] = r < ctx->nRanks ? W->peer[r] : nullptr;
    out->redOff = W->redOffK / sizeof(uint4);
    out->d_redSeq = W->d_redSeqK;
    out->me = ctx->rank;
    out->n = ctx->nRanks;
    return true;
}

void comm_peer_free_addr(ldu_addr* a)
{
    if (!a->peer) return;
    if (a->ctx->comm && a->ctx->comm->peer && a->peer->winBytes) a->ctx->comm->peer->release(a->peer->winOff, a->peer->winBytes);
    if (a->ctx->comm && a->ctx->comm->peer && a->peer->kWinBytes) a->ctx->comm->peer->release(a->peer->kWinOff, a->peer->kWinBytes);
    if (a->ctx->comm && a->ctx->comm->peer && a->peer->bWinBytes) a->ctx->comm->peer->release(a->peer->bWinOff, a->peer->bWinBytes);
    if (a->peer->d_bdst) (void)hipFree(a->peer->d_bdst);
    if (a->peer->d_bsrc) (void)hipFree((void*)a->peer->d_bsrc);
    if (a->peer->d_dst) (void)hipFree(a->peer->d_dst);
    if (a->peer->d_src) (void)hipFree((void*)a->peer->d_src);
    if (a->peer->d_kdst) (void)hipFree(a->peer->d_kdst);
    if (a->peer->d_ksrc) (void)hipFree((void*)a->peer->d_ksrc);
    if (a->peer->d_kseq) (void)hipFree(a->peer->d_kseq);
    delete a->peer;
    a->peer = nullptr;
}

void comm_destroy(ldu_ctx* ctx)
{
    if (!ctx->comm) return;
    if (ctx->comm->comm) ncclCommDestroy(ctx->comm->comm);
    if (ctx->comm->d_ibuf) (void)hipFree(ctx->comm->d_ibuf);
    if (ctx->comm->d_gbuf) (void)hipFree(ctx->comm->d_gbuf);
    if (PeerWindow* W = ctx->comm->peer)
    {
        for (size_t r = 0; r < W->peer.size(); r++)
            if (W->opened[r] && W->peer[r]) (void)hipIpcCloseMemHandle(W->peer[r]);
        if (W->base) (void)hipFree(W->base);
        if (W->d_redSeqK) (void)hipFree(W->d_redSeqK);
        delete W;
    }
    delete ctx->comm;
    ctx->comm = nullptr;
}

int comm_allreduce_scalars(ldu_ctx* ctx, int slot, int count, hipStream_t s)
{
    // LDU_FORCE_COMM=1 sends even a 1-rank reduction through RCCL (exercises the backend on 1 GPU)
    const char* fe = getenv("LDU_FORCE_COMM");   // (read per call: tests switch it within one process)
    const bool force = fe && atoi(fe);
    if (!ctx->comm || (ctx->nRanks <= 1 && !force)) return 0;
    ctx->nAllReduces++;
    if (ctx->comm->peerReduce)
    {
        PeerWindow* W = ctx->comm->peer;
        PeerRed P;
        for (int r = 0; r < LDU_MAX_PEERS; r++) P.win[r] = r < ctx->nRanks ? W->peer[r] : nullptr;
        return k_peer_allreduce(ctx, P, W->redOff / sizeof(uint4), ctx->rank, ctx->nRanks, count, ++W->redSeq, ctx->S() + slot,
                                nullptr, s);
    }
    if (ctx->comm->local)
    {
        LocalGroup* G = ctx->comm->local;
        LDU_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<double>& mine = G->dbuf[ctx->rank];
        LDU_CHECK_HIP(hipMemcpy(mine.data(), ctx->S() + slot, sizeof(double) * count, hipMemcpyDeviceToHost));
        G->barrier();
        double sum[S_BANK];
        for (int i = 0; i < count; i++)
        {
            double t = G->dbuf[0][i];
            for (int r = 1; r < G->n; r++) t += G->dbuf[r][i];   // rank order, like the oracle
            sum[i] = t;
        }
        G->barrier();
        LDU_CHECK_HIP(hipMemcpy(ctx->S() + slot, sum, sizeof(double) * count, hipMemcpyHostToDevice));
        return 0;
    }
    LDU_CHECK_NCCL(ncclAllReduce(ctx->S() + slot, ctx->S() + slot, (size_t)count, ncclDouble, ncclSum,
                                 ctx->comm->comm, s));
    return 0;
}

// The abort flag of the bounded dependency waits, made collective: max over the ranks, on the stream, right before
// a rank reads it.  Every rank issues the same sequence of exchanges and reductions between two reads whether or not
// one of its sweeps gave up (an aborted sweep drains, the operation carries on), so with this every rank takes the
// engine fallback at the same point of the same operation, or none does (run_with_fallback).
int comm_allreduce_abort(ldu_ctx* ctx, hipStream_t s)
{
    const char* fe = getenv("LDU_FORCE_COMM");   // (read per call: tests switch it within one process)
    const bool force = fe && atoi(fe);
    if (!ctx->comm || (ctx->nRanks <= 1 && !force)) return 0;
    if (ctx->comm->peerReduce)
    {
        PeerWindow* W = ctx->comm->peer;
        PeerRed P;
        for (int r = 0; r < LDU_MAX_PEERS; r++) P.win[r] = r < ctx->nRanks ? W->peer[r] : nullptr;
        // (both words: the sweep engines' abort flag and the singular-matrix flag of directSolveCoarsest - every rank takes
        //  the same error path, none runs on into the next collective alone)
        return k_peer_allreduce(ctx, P, W->redOff / sizeof(uint4), ctx->rank, ctx->nRanks, 2, ++W->redSeq, nullptr,
                                ctx->d_abort, s);
    }
    if (ctx->comm->local)
    {
        LocalGroup* G = ctx->comm->local;
        int mine[2] = {0, 0};
        LDU_CHECK_HIP(hipStreamSynchronize(s));
        LDU_CHECK_HIP(hipMemcpy(mine, ctx->d_abort, 2 * sizeof(int), hipMemcpyDeviceToHost));
        G->dbuf[ctx->rank][0] = (double)mine[0];
        G->dbuf[ctx->rank][1] = (double)mine[1];
        G->barrier();
        int any[2] = {0, 0};
        for (int r = 0; r < G->n; r++) { any[0] |= G->dbuf[r][0] != 0.0; any[1] |= G->dbuf[r][1] != 0.0; }
        G->barrier();
        if ((any[0] && !mine[0]) || (any[1] && !mine[1]))
            LDU_CHECK_HIP(hipMemcpy(ctx->d_abort, any, 2 * sizeof(int), hipMemcpyHostToDevice));
        return 0;
    }
    LDU_CHECK_NCCL(ncclAllReduce(ctx->d_abort, ctx->d_abort, 2, ncclInt, ncclMax, ctx->comm->comm, s));
    return 0;
}

// index of the patch on rank `nbr` that pairs with my patch `p`: the k-th patch of nbr towards me,
// k = ordinal of p among my patches towards nbr
static int paired_patch(const std::vector<Patch>& mine, int p, const std::vector<Patch>& theirs, int me)
{
    int k = 0;
    for (int i = 0; i < p; i++) if (mine[i].nbrRank == mine[p].nbrRank) k++;
    for (int j = 0; j < (int)theirs.size(); j++)
        if (theirs[j].nbrRank == me && k-- == 0) return j;
    return -1;
}

// the patches comm_exchange sends / receives for, in the order it issues the ncclSend / ncclRecv pairs (processor patches
// with faces; cyclic patches are local copies).  RCCL matches the k-th send to a peer with the k-th receive posted for
// it there, so this order together with paired_patch IS the wire protocol.
static std::vector<int> comm_remote_order(const std::vector<Patch>& patches)
{
    std::vector<int> order;
    for (int p = 0; p < (int)patches.size(); p++)
        if (patches[p].n != 0 && patches[p].nbrPatch < 0) order.push_back(p);
    return order;
}

// Host-only views of the two rules above (no device, no communicator): the CPU tests drive them over gloo
// (tests/test_gloo_2rank.py) so that a change of the pairing or of the issue order fails there.
extern "C" int ldu_comm_paired_patch(int32_t nMine, const int32_t* mineNbrRank, int32_t p, int32_t nTheirs,
                                     const int32_t* theirsNbrRank, int32_t me)
{
    if (nMine < 0 || nTheirs < 0 || p < 0 || p >= nMine || !mineNbrRank || (nTheirs && !theirsNbrRank)) return -2;
    std::vector<Patch> mine(nMine), theirs(nTheirs);
    for (int i = 0; i < nMine; i++) mine[i].nbrRank = mineNbrRank[i];
    for (int i = 0; i < nTheirs; i++) theirs[i].nbrRank = theirsNbrRank[i];
    return paired_patch(mine, p, theirs, me);
}

extern "C" int ldu_comm_exchange_order(int32_t nPatches, const int32_t* nFaces, const int32_t* nbrPatch, int32_t* order)
{
    if (nPatches < 0 || (nPatches && (!nFaces || !nbrPatch || !order))) return -2;
    std::vector<Patch> patches(nPatches);
    for (int i = 0; i < nPatches; i++) { patches[i].n = nFaces[i]; patches[i].nbrPatch = nbrPatch[i]; }
    const std::vector<int> o = comm_remote_order(patches);
    for (size_t i = 0; i < o.size(); i++) order[i] = o[i];
    return (int)o.size();
}

bool comm_is_peer(const ldu_ctx* ctx) { return ctx->comm && ctx->comm->peerHalo; }

// initMatrixInterfaces (lduMatrixUpdateMatrixInterfaces.C:30-93): pack the coupled faces' cell values and start the exchange
int comm_halo_pack_exchange(ldu_addr* a, const double* x, hipStream_t s)
{
    if (!a->nPatchFaces) return 0;
    if (a->peer && a->ctx->comm->peerHalo)
    {
        // peer stores: the pack kernel IS the send (ldu_peer.hip); the receive is polled in comm_wait_halo
        ldu_ctx* ctx = a->ctx;
        if (a->peer->pending) { ldu_set_error("halo exchange started twice without an update in between"); return -4; }
        if (k_peer_pack(a, x, ++a->peer->seq, s)) return -1;
        for (auto& P : a->patches)
            if (P.nbrPatch >= 0 && P.n)
                LDU_CHECK_HIP(hipMemcpyAsync(P.d_recv, a->patches[P.nbrPatch].d_send, sizeof(double) * P.n,
                                             hipMemcpyDeviceToDevice, s));
        a->peer->pending = true;
        ctx->nHaloExchanges++;
        ctx->nHaloOverlapped++;   // the interior rows run between pack and unpack on the same stream
        return 0;
    }
    if (k_pack_patches(a, x, s)) return -1;
    return comm_exchange(a, s);
}

int comm_exchange(ldu_addr* a, hipStream_t s)
{
    ldu_ctx* ctx = a->ctx;
    if (a->patches.empty()) return 0;
    // cyclic patches: the neighbour values are this rank's own send buffer of the paired patch
    // (cyclicGAMGInterface::internalFieldTransfer / cyclicFvPatchField::patchNeighbourField)
    bool remote = false;
    for (auto& P : a->patches)
    {
        if (P.nbrPatch < 0) { remote = true; continue; }
        if (P.n)
            LDU_CHECK_HIP(hipMemcpyAsync(P.d_recv, a->patches[P.nbrPatch].d_send, sizeof(double) * P.n,
                                         hipMemcpyDeviceToDevice, s));
    }
    if (!remote) return 0;
    if (!ctx->comm)
    {
        ldu_set_error("processor patches present but no communicator (ldu_ctx_comm_init)");
        return -4;
    }
    ctx->nHaloExchanges++;
    if (ctx->comm->local)
    {
        LocalGroup* G = ctx->comm->local;
        LDU_CHECK_HIP(hipStreamSynchronize(s));   // my send buffers are packed
        G->addr[ctx->rank] = a;
        G->barrier();
        for (int p = 0; p < (int)a->patches.size(); p++)
        {
            Patch& P = a->patches[p];
            if (!P.n || P.nbrPatch >= 0) continue;
            const ldu_addr* na = G->addr[P.nbrRank];
            const int q = paired_patch(a->patches, p, na->patches, ctx->rank);
            if (q < 0 || na->patches[q].n != P.n) { ldu_set_error("local exchange: unpaired patch"); return -4; }
            LDU_CHECK_HIP(hipMemcpy(P.d_recv, na->patches[q].d_send, sizeof(double) * P.n, hipMemcpyDeviceToDevice));
        }
        LDU_CHECK_HIP(hipDeviceSynchronize());
        G->barrier();
        return 0;
    }
    // the exchange runs on the comm stream: it starts when the send buffers are packed (event on the compute stream)
    // and the compute stream only waits for it where the received values are consumed (comm_wait_halo, before
    // apply_patches) - the interior rows of Amul / residual run meanwhile
    hipStream_t cs = s;
    if (ctx->haloOverlap)
    {
        if (ctx->haloInFlight) LDU_CHECK_HIP(hipStreamWaitEvent(s, ctx->evHalo, 0));   // never two exchanges in flight
        LDU_CHECK_HIP(hipEventRecord(ctx->evPacked, s));
        LDU_CHECK_HIP(hipStreamWaitEvent(ctx->streamComm, ctx->evPacked, 0));
        cs = ctx->streamComm;
    }
    LDU_CHECK_NCCL(ncclGroupStart());
    for (int pi : comm_remote_order(a->patches))
    {
        Patch& p = a->patches[pi];
        LDU_CHECK_NCCL(ncclSend(p.d_send, (size_t)p.n, ncclDouble, p.nbrRank, ctx->comm->comm, cs));
        LDU_CHECK_NCCL(ncclRecv(p.d_recv, (size_t)p.n, ncclDouble, p.nbrRank, ctx->comm->comm, cs));
    }
    LDU_CHECK_NCCL(ncclGroupEnd());
    if (ctx->haloOverlap)
    {
        LDU_CHECK_HIP(hipEventRecord(ctx->evHalo, cs));
        ctx->haloInFlight = true;
        ctx->nHaloOverlapped++;
    }
    return 0;
}

// the compute stream waits for the exchange started by the last comm_exchange (no-op when none is in flight)
int comm_wait_halo(ldu_addr* a, hipStream_t s)
{
    ldu_ctx* ctx = a->ctx;
    if (a->peer && a->peer->pending)
    {
        a->peer->pending = false;
        return k_peer_unpack(a, a->peer->seq, s);
    }
    if (!ctx->haloInFlight) return 0;
    LDU_CHECK_HIP(hipStreamWaitEvent(s, ctx->evHalo, 0));
    if (s == ctx->stream) ctx->haloInFlight = false;
    return 0;
}

// setup-time: min-reduce of one int over the ranks (continueAgglomerating, GAMGAgglomeration.C:53-62)
int comm_allreduce_min_int(ldu_ctx* ctx, int* v)
{
    if (!ctx->comm || ctx->nRanks <= 1) return 0;
    if (ctx->comm->peer && !ctx->comm->comm)
    {
        // out-of-band: every rank sends its value to every other rank
        const int n = ctx->nRanks;
        std::vector<int> peers, vals(n, 0), mine(n, *v);
        std::vector<const void*> sp;
        std::vector<void*> rp;
        std::vector<int64_t> nb;
        for (int r = 0; r < n; r++)
            if (r != ctx->rank) { peers.push_back(r); sp.push_back(&mine[r]); rp.push_back(&vals[r]); nb.push_back(sizeof(int)); }
        if (peer_oob(ctx, peers, sp, nb, rp, nb)) return -1;
        for (int r = 0; r < n; r++) if (r != ctx->rank) *v = std::min(*v, vals[r]);
        return 0;
    }
    if (ctx->comm->local)
    {
        LocalGroup* G = ctx->comm->local;
        G->dbuf[ctx->rank][0] = (double)*v;
        G->barrier();
        double m = G->dbuf[0][0];
        for (int r = 1; r < G->n; r++) m = std::min(m, G->dbuf[r][0]);
        G->barrier();
        *v = (int)m;
        return 0;
    }
    hipStream_t s = ctx->stream;
    int* d = nullptr;
    LDU_CHECK_HIP(hipMalloc((void**)&d, sizeof(int)));
    LDU_CHECK_HIP(hipMemcpyAsync(d, v, sizeof(int), hipMemcpyHostToDevice, s));
    LDU_CHECK_NCCL(ncclAllReduce(d, d, 1, ncclInt, ncclMin, ctx->comm->comm, s));
    LDU_CHECK_HIP(hipMemcpyAsync(v, d, sizeof(int), hipMemcpyDeviceToHost, s));
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    (void)hipFree(d);
    return 0;
}

// all-gather of one host buffer per rank (sizes may differ): all[r] = rank r's bytes, on every rank.  Host-staged on every
// carrier - used where the reference itself gathers on the master through Pstream (the coarsest-level matrices and sources of
// directSolveCoarsest, LUscalarMatrix.C:52-107, LUscalarMatrixTemplates.C:31-118); not a hot path.
int comm_allgather_host(ldu_ctx* ctx, const void* mine, int64_t nBytes, std::vector<std::vector<char>>& all)
{
    const int n = (ctx->comm && ctx->nRanks > 1) ? ctx->nRanks : 1;
    const int me = n > 1 ? ctx->rank : 0;
    all.assign(n, std::vector<char>());
    all[me].assign((const char*)mine, (const char*)mine + nBytes);
    if (n == 1) return 0;
    if (ctx->comm->peer && !ctx->comm->comm)
    {
        // out-of-band: sizes first, then every rank's bytes to every other rank
        std::vector<int> peers;
        std::vector<int64_t> sizes(n, 0), mineSz(n, nBytes), eight;
        std::vector<const void*> sp;
        std::vector<void*> rp;
        for (int r = 0; r < n; r++)
            if (r != me) { peers.push_back(r); sp.push_back(&mineSz[r]); rp.push_back(&sizes[r]); eight.push_back(sizeof(int64_t)); }
        if (peer_oob(ctx, peers, sp, eight, rp, eight)) return -1;
        std::vector<int64_t> sb, rb;
        sp.clear(); rp.clear();
        for (int r = 0; r < n; r++)
            if (r != me)
            {
                all[r].resize((size_t)sizes[r]);
                sp.push_back(all[me].data()); sb.push_back(nBytes);
                rp.push_back(all[r].data()); rb.push_back(sizes[r]);
            }
        return peer_oob(ctx, peers, sp, sb, rp, rb) ? -1 : 0;
    }
    if (ctx->comm->local)
    {
        LocalGroup* G = ctx->comm->local;
        static std::mutex mu;
        static std::map<std::pair<LocalGroup*, int>, const std::vector<char>*> pubs;
        {
            std::lock_guard<std::mutex> lk(mu);
            pubs[{G, me}] = &all[me];
        }
        G->barrier();
        for (int r = 0; r < n; r++)
            if (r != me)
            {
                std::lock_guard<std::mutex> lk(mu);
                all[r] = *pubs[{G, r}];
            }
        G->barrier();
        return 0;
    }
    // RCCL: the sizes, then the buffers padded to the largest
    hipStream_t s = ctx->stream;
    ldu_comm_impl* C = ctx->comm;
    auto stage = [&](size_t bytes) -> int {
        if (C->gbufCap >= bytes) return 0;
        if (C->d_gbuf) (void)hipFree(C->d_gbuf);
        C->d_gbuf = nullptr; C->gbufCap = 0;
        LDU_CHECK_HIP(hipMalloc((void**)&C->d_gbuf, bytes));
        C->gbufCap = bytes;
        return 0;
    };
    if (stage(sizeof(int64_t) * (size_t)(n + 1))) return -1;
    std::vector<int64_t> sizes(n, 0);
    LDU_CHECK_HIP(hipMemcpyAsync(C->d_gbuf, &nBytes, sizeof(int64_t), hipMemcpyHostToDevice, s));
    LDU_CHECK_NCCL(ncclAllGather(C->d_gbuf, C->d_gbuf + sizeof(int64_t), 1, ncclInt64, C->comm, s));
    LDU_CHECK_HIP(hipMemcpyAsync(sizes.data(), C->d_gbuf + sizeof(int64_t), sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    size_t mx = 1;
    for (int r = 0; r < n; r++) mx = std::max(mx, (size_t)sizes[r]);
    mx = (mx + 15) & ~(size_t)15;
    if (stage(mx * (size_t)(n + 1))) return -1;
    if (nBytes) LDU_CHECK_HIP(hipMemcpyAsync(C->d_gbuf, mine, (size_t)nBytes, hipMemcpyHostToDevice, s));
    LDU_CHECK_NCCL(ncclAllGather(C->d_gbuf, C->d_gbuf + mx, mx, ncclChar, C->comm, s));
    for (int r = 0; r < n; r++)
        if (r != me)
        {
            all[r].resize((size_t)sizes[r]);
            if (sizes[r])
                LDU_CHECK_HIP(hipMemcpyAsync(all[r].data(), C->d_gbuf + mx * (size_t)(r + 1), (size_t)sizes[r], hipMemcpyDeviceToHost, s));
        }
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

// setup-time: per-patch exchange of int labels with the neighbour rank
// (processorGAMGInterface::initInternalFieldTransfer / internalFieldTransfer, :137-154)
int comm_exchange_ints(ldu_ctx* ctx, const std::vector<Patch>& patches,
                       const std::vector<std::vector<int>>& send, std::vector<std::vector<int>>& recv)
{
    recv.assign(patches.size(), std::vector<int>());
    if (patches.empty()) return 0;
    bool remote = false;
    for (size_t p = 0; p < patches.size(); p++)
    {
        if (patches[p].nbrPatch >= 0) recv[p] = send[patches[p].nbrPatch];   // cyclic: local
        else remote = true;
    }
    if (!remote) return 0;
    if (!ctx->comm) { ldu_set_error("processor patches present but no communicator"); return -4; }
    if (ctx->comm->peer && !ctx->comm->comm && ctx->nRanks == 1)
    {
        // one rank whose processor patches face each other (bench.py --rank-of with the peer carrier alone): the labels of the
        // paired patch (itself when alone), as peer_addr_setup pairs the windows
        for (int p = 0; p < (int)patches.size(); p++)
            if (patches[p].nbrPatch < 0)
            {
                const int q = paired_patch(patches, p, patches, ctx->rank);
                recv[p] = send[q < 0 ? p : q];
            }
        return 0;
    }
    if (ctx->comm->peer && !ctx->comm->comm)
    {
        // out-of-band, one message per neighbour rank: the int lists of all patches towards it, in patch order - the
        // neighbour lists its patches towards me in the paired order (paired_patch: k-th with k-th), equal sizes
        std::map<int, std::vector<int>> byRank;
        for (int p = 0; p < (int)patches.size(); p++)
            if (patches[p].nbrPatch < 0 && patches[p].n > 0) byRank[patches[p].nbrRank].push_back(p);
        std::vector<int> peers;
        std::vector<std::vector<int>> sb, rb;
        for (auto& kv : byRank)
        {
            std::vector<int> buf;
            for (int p : kv.second) buf.insert(buf.end(), send[p].begin(), send[p].end());
            peers.push_back(kv.first);
            rb.emplace_back(buf.size());
            sb.push_back(std::move(buf));
        }
        std::vector<const void*> sp;
        std::vector<void*> rp;
        std::vector<int64_t> nb;
        for (size_t i = 0; i < peers.size(); i++)
        {
            sp.push_back(sb[i].data()); rp.push_back(rb[i].data()); nb.push_back((int64_t)(sizeof(int) * sb[i].size()));
        }
        if (peer_oob(ctx, peers, sp, nb, rp, nb)) return -1;
        size_t i = 0;
        for (auto& kv : byRank)
        {
            size_t off = 0;
            for (int p : kv.second)
            {
                recv[p].assign(rb[i].begin() + off, rb[i].begin() + off + send[p].size());
                off += send[p].size();
            }
            i++;
        }
        return 0;
    }
    if (ctx->comm->local)
    {
        LocalGroup* G = ctx->comm->local;
        // pairing needs the neighbours' patch lists: publish (patches, send) of this rank
        static thread_local std::vector<Patch> dummy;
        struct Pub { const std::vector<Patch>* patches; const std::vector<std::vector<int>>* send; };
        static std::mutex mu;
        static std::map<std::pair<LocalGroup*, int>, Pub> pubs;
        {
            std::lock_guard<std::mutex> lk(mu);
            pubs[{G, ctx->rank}] = Pub{&patches, &send};
        }
        G->barrier();
        for (int p = 0; p < (int)patches.size(); p++)
        {
            if (patches[p].nbrPatch >= 0) continue;
            Pub nb;
            {
                std::lock_guard<std::mutex> lk(mu);
                nb = pubs[{G, patches[p].nbrRank}];
            }
            const int q = paired_patch(patches, p, *nb.patches, ctx->rank);
            if (q < 0) { ldu_set_error("local int exchange: unpaired patch"); return -4; }
            recv[p] = (*nb.send)[q];
        }
        G->barrier();
        return 0;
    }
    hipStream_t s = ctx->stream;
    size_t tot = 0;
    for (auto& v : send) tot += v.size();
    if (ctx->comm->ibufCap < 2 * tot + 2)
    {
        if (ctx->comm->d_ibuf) (void)hipFree(ctx->comm->d_ibuf);
        ctx->comm->ibufCap = 2 * tot + 2;
        LDU_CHECK_HIP(hipMalloc((void**)&ctx->comm->d_ibuf, sizeof(int) * ctx->comm->ibufCap));
    }
    int* ds = ctx->comm->d_ibuf;
    int* dr = ctx->comm->d_ibuf + tot;
    size_t off = 0;
    for (auto& v : send)
    {
        if (v.size()) LDU_CHECK_HIP(hipMemcpyAsync(ds + off, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice, s));
        off += v.size();
    }
    LDU_CHECK_NCCL(ncclGroupStart());
    off = 0;
    for (size_t p = 0; p < patches.size(); p++)
    {
        const size_t n = send[p].size();
        if (n && patches[p].nbrPatch < 0)
        {
            LDU_CHECK_NCCL(ncclSend(ds + off, n, ncclInt, patches[p].nbrRank, ctx->comm->comm, s));
            LDU_CHECK_NCCL(ncclRecv(dr + off, n, ncclInt, patches[p].nbrRank, ctx->comm->comm, s));
        }
        off += n;
    }
    LDU_CHECK_NCCL(ncclGroupEnd());
    off = 0;
    for (size_t p = 0; p < patches.size(); p++)
    {
        if (patches[p].nbrPatch < 0) recv[p].resize(send[p].size());
        if (send[p].size() && patches[p].nbrPatch < 0)
            LDU_CHECK_HIP(hipMemcpyAsync(recv[p].data(), dr + off, sizeof(int) * send[p].size(), hipMemcpyDeviceToHost, s));
        off += send[p].size();
    }
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}
