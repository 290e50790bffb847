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

struct ldu_comm_impl {
    ncclComm_t comm = nullptr;
    LocalGroup* local = nullptr;
    int* d_ibuf = nullptr;       // staging for int exchanges (RCCL)
    size_t ibufCap = 0;
};

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
    ctx->comm = new ldu_comm_impl();
    LDU_CHECK_NCCL(ncclCommInitRank(&ctx->comm->comm, nRanks, u, rank));
    ctx->rank = rank;
    ctx->nRanks = nRanks;
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

void comm_destroy(ldu_ctx* ctx)
{
    if (!ctx->comm) return;
    if (ctx->comm->comm) ncclCommDestroy(ctx->comm->comm);
    if (ctx->comm->d_ibuf) (void)hipFree(ctx->comm->d_ibuf);
    delete ctx->comm;
    ctx->comm = nullptr;
}

int comm_allreduce_scalars(ldu_ctx* ctx, int slot, int count, hipStream_t s)
{
    // LDU_FORCE_COMM=1 sends even a 1-rank reduction through RCCL (exercises the backend on 1 GPU)
    static const bool force = getenv("LDU_FORCE_COMM") && atoi(getenv("LDU_FORCE_COMM"));
    if (!ctx->comm || (ctx->nRanks <= 1 && !force)) return 0;
    ctx->nAllReduces++;
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
    static const bool force = getenv("LDU_FORCE_COMM") && atoi(getenv("LDU_FORCE_COMM"));
    if (!ctx->comm || (ctx->nRanks <= 1 && !force)) return 0;
    if (ctx->comm->local)
    {
        LocalGroup* G = ctx->comm->local;
        int mine = 0;
        LDU_CHECK_HIP(hipStreamSynchronize(s));
        LDU_CHECK_HIP(hipMemcpy(&mine, ctx->d_abort, sizeof(int), hipMemcpyDeviceToHost));
        G->dbuf[ctx->rank][0] = (double)mine;
        G->barrier();
        int any = 0;
        for (int r = 0; r < G->n; r++) any |= G->dbuf[r][0] != 0.0;
        G->barrier();
        if (any && !mine) LDU_CHECK_HIP(hipMemcpy(ctx->d_abort, &any, sizeof(int), hipMemcpyHostToDevice));
        return 0;
    }
    LDU_CHECK_NCCL(ncclAllReduce(ctx->d_abort, ctx->d_abort, 1, ncclInt, ncclMax, ctx->comm->comm, s));
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
int comm_wait_halo(ldu_ctx* ctx, hipStream_t s)
{
    if (!ctx->haloInFlight) return 0;
    LDU_CHECK_HIP(hipStreamWaitEvent(s, ctx->evHalo, 0));
    if (s == ctx->stream) ctx->haloInFlight = false;
    return 0;
}

// setup-time: min-reduce of one int over the ranks (continueAgglomerating, GAMGAgglomeration.C:53-62)
int comm_allreduce_min_int(ldu_ctx* ctx, int* v)
{
    if (!ctx->comm || ctx->nRanks <= 1) return 0;
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
