// Communication layer: RCCL over xGMI, one rank per GPU / sub-domain.
// Replaces the MPI calls reached from the hot path:
//   reduce(scalar, sumOp) -> ncclAllReduce on device-resident scalars (never leaves the stream);
//   processor-patch Isend/Irecv -> grouped ncclSend/ncclRecv of the packed halo buffers.
// Both are enqueued on the compute stream, so they are ordered with the kernels without any
// host synchronisation; interior rows overlap the exchange on the second stream where used.
#include <rccl/rccl.h>

#include <cstring>

#include "ldu_internal.hpp"

struct ldu_comm_impl {
    ncclComm_t comm;
};

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

void comm_destroy(ldu_ctx* ctx)
{
    if (ctx->comm)
    {
        ncclCommDestroy(ctx->comm->comm);
        delete ctx->comm;
        ctx->comm = nullptr;
    }
}

int comm_allreduce_scalars(ldu_ctx* ctx, int slot, int count, hipStream_t s)
{
    if (!ctx->comm || ctx->nRanks <= 1) return 0;
    LDU_CHECK_NCCL(ncclAllReduce(ctx->S() + slot, ctx->S() + slot, (size_t)count, ncclDouble,
                                 ncclSum, ctx->comm->comm, s));
    return 0;
}

int comm_exchange(ldu_addr* a, hipStream_t s)
{
    ldu_ctx* ctx = a->ctx;
    if (a->patches.empty()) return 0;
    if (!ctx->comm)
    {
        ldu_set_error("coupled patches present but no communicator (ldu_ctx_comm_init)");
        return -4;
    }
    LDU_CHECK_NCCL(ncclGroupStart());
    for (auto& p : a->patches)
    {
        if (p.n == 0) continue;
        LDU_CHECK_NCCL(ncclSend(p.d_send, (size_t)p.n, ncclDouble, p.nbrRank, ctx->comm->comm, s));
        LDU_CHECK_NCCL(ncclRecv(p.d_recv, (size_t)p.n, ncclDouble, p.nbrRank, ctx->comm->comm, s));
    }
    LDU_CHECK_NCCL(ncclGroupEnd());
    return 0;
}
