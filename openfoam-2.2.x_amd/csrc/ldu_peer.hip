// Peer-store communication kernels (backend "peer" of ldu_comm.cpp): halo exchange and global sums written straight
// into the neighbour GPU's memory over xGMI, no collective library in the steady state.
//
// What they replace in the reference: the processor-patch Isend / Irecv of
// lduMatrix::initMatrixInterfaces / updateMatrixInterfaces (lduMatrixUpdateMatrixInterfaces.C:30-160,
// processorFvPatchField.C:375-450) and reduce(scalar, sumOp) of gSumProd / gSumMag (FieldFunctions.C:514-533).
//
// Mechanism: every rank owns a WINDOW of fine-grained device memory that all other ranks of the node have mapped
// (hipIpc, or the plain pointer inside one process).  A value travels as a 16-byte granule {v_lo, tag, v_hi, tag}
// written with ONE system-scope store; the consumer polls the granule in its OWN window with system-scope loads until
// both tags equal the expected sequence number - the data is the flag, each 8-byte half validates itself, no fence,
// no reset pass (the same granule the point-to-point sweep engines use inside one GPU, ldu_kernels.hip).  Buffers are
// double-buffered by the parity of the sequence number: a pairwise exchange cannot be overtaken by more than one
// (the sender's exchange k+2 follows its own unpack k+1, which saw the receiver's pack k+1, which the receiver issued
// after its unpack k), so parity k is free again when exchange k+2 writes it.
//
// Measured on one MI355X with 2 ... 4 PROCESSES sharing the GPU (tools/ipc_probe.hip): all-reduce of 4 doubles
// 3.1-4.2 us per operation back to back, halo pack + unpack 6 us per exchange independent of the patch size up to
// 46 656 faces, 1.2 us per ping-pong round trip between two running kernels.
#include "ldu_peer_dev.hpp"

// initMatrixInterfaces: send[i] = x[faceCells[i]] for every coupled face (cyclic patches read d_send), and for the
// faces of processor patches the same value as a granule into the neighbour's receive region
__global__ void peer_pack_kernel(int n, const int* __restrict__ pfCell, const double* __restrict__ x,
                                 double* __restrict__ send, uint4* const* __restrict__ dst, unsigned seq)
{
    for (int i = blockIdx.x * PBLK + threadIdx.x; i < n; i += gridDim.x * PBLK)
    {
        const double v = x[pfCell[i]];
        send[i] = v;
        uint4* d = dst[i];
        if (d) peer_store(d, v, seq);
    }
}

// updateMatrixInterfaces, first half: wait for the neighbour's values of exchange `seq`, leave them in recv[]
__global__ void peer_unpack_kernel(int n, const uint4* const* __restrict__ src, double* __restrict__ recv, unsigned seq,
                                   int* abortFlag)
{
    for (int i = blockIdx.x * PBLK + threadIdx.x; i < n; i += gridDim.x * PBLK)
    {
        const uint4* s = src[i];
        if (!s) continue;
        double v = 0.0;
        unsigned spins = 0;
        unsigned long long tw0 = 0;
        bool ok = true;
        while (!peer_load(s, seq, v))
        {
            if (peer_wait_expired(spins, tw0, abortFlag)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { abortFlag[LDU_PEER_FLAG] = 1; v = 0.0; }
        recv[i] = v;
    }
}

int k_peer_pack(ldu_addr* a, const double* x, unsigned seq, hipStream_t s)
{
    const int n = a->nPatchFaces;
    const int grid = std::min((n + PBLK - 1) / PBLK, 1024);
    peer_pack_kernel<<<grid, PBLK, 0, s>>>(n, a->d_pfCell, x, a->d_sendAll, a->peer->d_dst + (size_t)(seq & 1u) * n, seq);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

int k_peer_unpack(ldu_addr* a, unsigned seq, hipStream_t s)
{
    const int n = a->nPatchFaces;
    const int grid = std::min((n + PBLK - 1) / PBLK, 1024);
    peer_unpack_kernel<<<grid, PBLK, 0, s>>>(n, a->peer->d_src + (size_t)(seq & 1u) * n, a->d_recvAll, seq, a->ctx->d_abort);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// Global sum of `count` (<= 16) doubles over n (<= 16) ranks in ONE single-workgroup kernel per rank: lane (r, i) stores
// my value i into rank r's window, then polls what rank r stored into mine; the sum is formed in RANK ORDER by every
// rank alike (deterministic, identical on all ranks - what the convergence decisions need; the reference's
// reduce() adds in a tree of ranks, Pstream/gatherScatter: ~1e-16 apart).
// Region layout in every window: [parity][source rank][16] granules.
// abortWord != nullptr: the `count` integers there are max-reduced instead (the collective engine fallback: the sweep
// engines' abort flag and the singular-matrix flag behind it).
__global__ void __launch_bounds__(256) peer_allreduce_kernel(PeerRed P, size_t redOff, int me, int n, int count, unsigned seq,
                                                             double* __restrict__ vals, int* abortWord, int* abortFlag)
{
    __shared__ double v[LDU_MAX_PEERS][16];
    const int lane = threadIdx.x;
    const int r = lane >> 4, i = lane & 15;
    const size_t par = (size_t)(seq & 1u) * LDU_MAX_PEERS * 16;
    if (r < n && i < count)
    {
        const double mine = abortWord ? (double)abortWord[i] : vals[i];
        peer_store(P.win[r] + redOff + par + (size_t)me * 16 + i, mine, seq);
        double x = 0.0;
        unsigned spins = 0;
        unsigned long long tw0 = 0;
        const uint4* src = P.win[me] + redOff + par + (size_t)r * 16 + i;
        while (!peer_load(src, seq, x))
        {
            if (peer_wait_expired(spins, tw0, abortFlag)) { abortFlag[LDU_PEER_FLAG] = 1; x = 0.0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        v[r][i] = x;
    }
    __syncthreads();
    if (lane < count)
    {
        if (abortWord)
        {
            double m = v[0][lane];
            for (int q = 1; q < n; q++) m = m > v[q][lane] ? m : v[q][lane];
            if (m != 0.0) abortWord[lane] = 1;
        }
        else
        {
            double t = v[0][lane];
            for (int q = 1; q < n; q++) t += v[q][lane];
            vals[lane] = t;
        }
    }
}

int k_peer_allreduce(ldu_ctx* ctx, const PeerRed& P, size_t redOff, int me, int n, int count, unsigned seq, double* vals,
                     int* abortWord, hipStream_t s)
{
    if (n > LDU_MAX_PEERS || count > 16) { ldu_set_error("peer all-reduce: more than 16 ranks or 16 values"); return -1; }
    peer_allreduce_kernel<<<1, 256, 0, s>>>(P, redOff, me, n, count, seq, vals, abortWord, ctx->d_abort);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

int k_peer_set_timeout(double seconds)
{
    const unsigned long long t = (unsigned long long)(seconds * 1e8);
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_peer_budget), &t, sizeof(t)));
    if (k_set_peer_timeout_kernels(t)) return -1;
    return k_coarsest_set_peer_timeout(t);   // (one copy of the budget per translation unit)
}
