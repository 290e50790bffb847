// The coarsest GAMG level (GAMGSolverSolve.C:430-487: PCG+DIC or PBiCG+DILU on ~10-60 cells with the outer
// tolerances) as ONE single-wavefront kernel.  Through the general solver the 17-cell system of the 216^3
// hierarchy cost 0.24 ms per V-cycle - a dozen launches and one host round trip per Krylov iteration.  Here
// the matrix is staged in LDS and one lane runs the reference's loops literally (face loops in face / losort
// order, left-to-right sums: PCG.C:65-182, PBiCG.C:65-198, DICPreconditioner.C:57-123,
// DILUPreconditioner.C:57-185, lduMatrixSolver.C:179-197), so this level is now also bit-identical to the CPU
// reference (the general path sums by a tree).  Vectors come in and go out in the plan's numbering.
#include <cstring>

#include "ldu_peer_dev.hpp"

#define CO_MAXC 64
#define CO_MAXF 512

template <bool BI>
__global__ void __launch_bounds__(LDU_WAVE)
coarsest_krylov_kernel(int n, int nF, const int* __restrict__ gl, const int* __restrict__ gu,
                       const int* __restrict__ glosort, const double* __restrict__ gdiag,
                       const double* __restrict__ gupper, const double* __restrict__ glower,
                       const int* __restrict__ perm, double* __restrict__ psiNew, const double* __restrict__ srcNew,
                       double tolerance, double relTol, int maxIter)
{
    __shared__ int l[CO_MAXF], u[CO_MAXF], losort[CO_MAXF];
    __shared__ double upper[CO_MAXF], lower[CO_MAXF];
    __shared__ double diag[CO_MAXC], psi[CO_MAXC], b[CO_MAXC], pA[CO_MAXC], wA[CO_MAXC], rA[CO_MAXC], rD[CO_MAXC];
    __shared__ double pT[BI ? CO_MAXC : 1], wT[BI ? CO_MAXC : 1], rT[BI ? CO_MAXC : 1];
    const int lane = threadIdx.x;
    for (int f = lane; f < nF; f += LDU_WAVE)
    {
        l[f] = gl[f]; u[f] = gu[f];
        upper[f] = gupper[f]; lower[f] = glower[f];
        if (BI) losort[f] = glosort[f];
    }
    for (int i = lane; i < n; i += LDU_WAVE)
    {
        const int o = perm[i];   // perm[new] = old
        psi[o] = psiNew[i];
        b[o] = srcNew[i];
    }
    for (int c = lane; c < n; c += LDU_WAVE) diag[c] = gdiag[c];
    __syncthreads();
    if (lane == 0)
    {
        const double great_ = 1e20, small_ = 1e-20, vsmall_ = 1e-300;
        // wA = A psi (wT = T psi), residuals
        for (int c = 0; c < n; c++) { wA[c] = diag[c] * psi[c]; if (BI) wT[c] = wA[c]; }
        for (int f = 0; f < nF; f++)
        {
            wA[u[f]] += lower[f] * psi[l[f]];
            wA[l[f]] += upper[f] * psi[u[f]];
            if (BI)
            {
                wT[u[f]] += upper[f] * psi[l[f]];
                wT[l[f]] += lower[f] * psi[u[f]];
            }
        }
        for (int c = 0; c < n; c++) { rA[c] = b[c] - wA[c]; if (BI) rT[c] = b[c] - wT[c]; }
        // normFactor (pA as the temporary, like the reference)
        for (int c = 0; c < n; c++) pA[c] = diag[c];
        for (int f = 0; f < nF; f++) { pA[u[f]] += lower[f]; pA[l[f]] += upper[f]; }
        double sum = 0.0;
        for (int c = 0; c < n; c++) sum += psi[c];
        const double avg = sum / (double)n;
        for (int c = 0; c < n; c++) pA[c] *= avg;
        double nf = 0.0;
        for (int c = 0; c < n; c++) nf += fabs(wA[c] - pA[c]) + fabs(b[c] - pA[c]);
        nf += small_;
        double res = 0.0;
        for (int c = 0; c < n; c++) res += fabs(rA[c]);
        const double initial = res / nf;
        double final_ = initial;
        bool converged = final_ < tolerance || (relTol > small_ && final_ < relTol * initial);
        if (!converged)
        {
            // calcReciprocalD
            for (int c = 0; c < n; c++) rD[c] = diag[c];
            for (int f = 0; f < nF; f++) rD[u[f]] -= upper[f] * lower[f] / rD[l[f]];
            for (int c = 0; c < n; c++) rD[c] = 1.0 / rD[c];
            if (BI) for (int c = 0; c < n; c++) pT[c] = 0.0;
            double wArA = great_, wArAold;
            int nIterations = 0;
            do
            {
                wArAold = wArA;
                // precondition (DIC: lower == upper; DILU: the lower sweep runs in losort order)
                for (int c = 0; c < n; c++) wA[c] = rD[c] * rA[c];
                for (int f = 0; f < nF; f++)
                {
                    const int sf = BI ? losort[f] : f;
                    wA[u[sf]] -= rD[u[sf]] * lower[sf] * wA[l[sf]];
                }
                for (int f = nF - 1; f >= 0; f--) wA[l[f]] -= rD[l[f]] * upper[f] * wA[u[f]];
                if (BI)
                {
                    for (int c = 0; c < n; c++) wT[c] = rD[c] * rT[c];
                    for (int f = 0; f < nF; f++) wT[u[f]] -= rD[u[f]] * upper[f] * wT[l[f]];
                    for (int f = nF - 1; f >= 0; f--)
                    {
                        const int sf = losort[f];
                        wT[l[sf]] -= rD[l[sf]] * lower[sf] * wT[u[sf]];
                    }
                }
                wArA = 0.0;
                for (int c = 0; c < n; c++) wArA += wA[c] * (BI ? rT[c] : rA[c]);
                if (nIterations == 0)
                {
                    for (int c = 0; c < n; c++) { pA[c] = wA[c]; if (BI) pT[c] = wT[c]; }
                }
                else
                {
                    const double beta = wArA / wArAold;
                    for (int c = 0; c < n; c++)
                    {
                        pA[c] = wA[c] + beta * pA[c];
                        if (BI) pT[c] = wT[c] + beta * pT[c];
                    }
                }
                for (int c = 0; c < n; c++) { wA[c] = diag[c] * pA[c]; if (BI) wT[c] = diag[c] * pT[c]; }
                for (int f = 0; f < nF; f++)
                {
                    wA[u[f]] += lower[f] * pA[l[f]];
                    wA[l[f]] += upper[f] * pA[u[f]];
                    if (BI)
                    {
                        wT[u[f]] += upper[f] * pT[l[f]];
                        wT[l[f]] += lower[f] * pT[u[f]];
                    }
                }
                double wApA = 0.0;
                for (int c = 0; c < n; c++) wApA += wA[c] * (BI ? pT[c] : pA[c]);
                if (fabs(wApA) / nf < vsmall_) break;   // checkSingularity
                const double alpha = wArA / wApA;
                for (int c = 0; c < n; c++)
                {
                    psi[c] += alpha * pA[c];
                    rA[c] -= alpha * wA[c];
                    if (BI) rT[c] -= alpha * wT[c];
                }
                res = 0.0;
                for (int c = 0; c < n; c++) res += fabs(rA[c]);
                final_ = res / nf;
                converged = final_ < tolerance || (relTol > small_ && final_ < relTol * initial);
            } while (nIterations++ < maxIter && !converged);
        }
    }
    __syncthreads();
    for (int i = lane; i < n; i += LDU_WAVE) psiNew[i] = psi[perm[i]];
}

// 1 = not taken (too large, coupled patches, several ranks, switched off)
int k_coarsest_solve(ldu_matrix* A, double tolerance, double relTol, int maxIter, double* corr, const double* src)
{
    ldu_addr* a = A->a;
    ldu_ctx* ctx = a->ctx;
    static const bool off = getenv("LDU_COARSEST_KERNEL") && !atoi(getenv("LDU_COARSEST_KERNEL"));
    if (off || a->nCells > CO_MAXC || a->nFaces > CO_MAXF || a->nCells == 0 || a->nFaces == 0 || a->nPatchFaces
        || ctx->nRanks > 1)
        return 1;
    hipStream_t s = ctx->stream;
    if (A->sym)
        coarsest_krylov_kernel<false><<<1, LDU_WAVE, 0, s>>>(a->nCells, a->nFaces, a->d_l, a->d_u, a->d_losort, A->d_diagO,
            A->d_upperO, A->d_upperO, a->d_perm, corr, src, tolerance, relTol, maxIter);
    else
        coarsest_krylov_kernel<true><<<1, LDU_WAVE, 0, s>>>(a->nCells, a->nFaces, a->d_l, a->d_u, a->d_losort, A->d_diagO,
            A->d_upperO, A->d_lowerO, a->d_perm, corr, src, tolerance, relTol, maxIter);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- the same across ranks, in ONE kernel per rank
// With processor patches (several ranks) the coarsest-level ICCG / BICCG of the reference is a distributed Krylov solve:
// per iteration one interface update inside Amul (two for BICCG: Amul and Tmul) and three global sums (wArA, wApA, the
// residual), ~10-40 cells per rank.  Through the general solver that is ~13 iterations x (a dozen launches, four
// collectives, one host read-back) per V-cycle - the largest single latency item of a multi-GPU V-cycle (VERDICT r3:
// 13 read-backs and ~50 of the ~140 collectives of a V-cycle).  With the peer-store backend a kernel can talk to the
// other ranks itself: every rank launches this kernel once, lane 0 runs the reference's loops on its sub-domain as
// above, and where the reference calls initMatrixInterfaces / updateMatrixInterfaces
// (lduMatrixUpdateMatrixInterfaces.C:30-160) or reduce() (FieldFunctions.C:514-533) the wavefront stores its patch
// values / partial sums into the neighbours' windows and polls its own (ldu_peer.hip: tagged granules, double-buffered
// by sequence parity; the sequence numbers live on the device because the iteration count is only known there).
// Sums are formed in rank order by every rank, so all ranks take the same convergence decisions - the arithmetic of the
// multi-domain oracle (oracle/ldu_oracle.c: rank-local loops, rank-ordered sums).  No launch, no read-back, no
// collective call inside the solve.
struct CoarsePeer {
    PeerKernelComm K;
    uint4* const* kdst;        // [2][nPF] (null table when the rank has no remote faces)
    const uint4* const* ksrc;
    unsigned* kseq;
    int nPF;
    const int* pfCellNew;      // [nPF] faceCells, plan numbering
    const int* cycPair;        // [nPF] cyclic faces: index of the paired face (own send value), -1 = remote
    const double* bou;         // [nPF]
    const double* intc;        // [nPF]
    int* abortFlag;
};

// all lanes: out[r][i] of every rank's vals[i]; returns the rank-ordered sum of value i in lane i (i < count <= 4)
__device__ __forceinline__ double co_allreduce(const CoarsePeer& C, unsigned& rseq, double mine, int count, double (*stage)[4])
{
    const int lane = threadIdx.x;
    const int r = lane >> 2, i = lane & 3;
    ++rseq;
    const size_t par = (size_t)(rseq & 1u) * LDU_MAX_PEERS * 16;
    const double v = __shfl(mine, i);     // lane i holds value i
    if (r < C.K.n && i < count)
    {
        peer_store(C.K.P.win[r] + C.K.redOff + par + (size_t)C.K.me * 16 + i, v, rseq);
        double x = 0.0;
        unsigned spins = 0;
        unsigned long long tw0 = 0;
        const uint4* src = C.K.P.win[C.K.me] + C.K.redOff + par + (size_t)r * 16 + i;
        while (!peer_load(src, rseq, x))
        {
            if (peer_wait_expired(spins, tw0, C.abortFlag)) { C.abortFlag[LDU_PEER_FLAG] = 1; x = 0.0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        stage[r][i] = x;
    }
    __syncthreads();
    double t = 0.0;
    if (lane < count)
    {
        t = stage[0][lane];
        for (int q = 1; q < C.K.n; q++) t += stage[q][lane];
    }
    __syncthreads();
    return t;
}

// all lanes: the neighbour values of vec at the coupled faces -> pnf[] (LDS)
__device__ __forceinline__ void co_halo(const CoarsePeer& C, unsigned& hseq, const double* vec, const int* pf, double* pnf)
{
    ++hseq;
    const size_t par = (size_t)(hseq & 1u) * C.nPF;
    for (int i = threadIdx.x; i < C.nPF; i += LDU_WAVE)
    {
        uint4* d = C.kdst ? C.kdst[par + i] : nullptr;
        if (d) peer_store(d, vec[pf[i]], hseq);
    }
    for (int i = threadIdx.x; i < C.nPF; i += LDU_WAVE)
    {
        const int cp = C.cycPair[i];
        double x = 0.0;
        if (cp >= 0) x = vec[pf[cp]];
        else
        {
            const uint4* s = C.ksrc[par + i];
            unsigned spins = 0;
            unsigned long long tw0 = 0;
            while (!peer_load(s, hseq, x))
            {
                if (peer_wait_expired(spins, tw0, C.abortFlag)) { C.abortFlag[LDU_PEER_FLAG] = 1; x = 0.0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        pnf[i] = x;
    }
    __syncthreads();
}

template <bool BI>
__global__ void __launch_bounds__(LDU_WAVE)
coarsest_krylov_peer_kernel(CoarsePeer C, int n, int nF, const int* __restrict__ gl, const int* __restrict__ gu,
                            const int* __restrict__ glosort, const double* __restrict__ gdiag,
                            const double* __restrict__ gupper, const double* __restrict__ glower,
                            const int* __restrict__ perm, double* __restrict__ psiNew, const double* __restrict__ srcNew,
                            double tolerance, double relTol, int maxIter)
{
    __shared__ int l[CO_MAXF], u[CO_MAXF], losort[CO_MAXF], pf[LDU_COARSEST_MAXP];
    __shared__ double upper[CO_MAXF], lower[CO_MAXF], bou[LDU_COARSEST_MAXP], intc[LDU_COARSEST_MAXP], pnf[LDU_COARSEST_MAXP];
    __shared__ double diag[CO_MAXC], psi[CO_MAXC], b[CO_MAXC], pA[CO_MAXC], wA[CO_MAXC], rA[CO_MAXC], rD[CO_MAXC];
    __shared__ double pT[BI ? CO_MAXC : 1], wT[BI ? CO_MAXC : 1], rT[BI ? CO_MAXC : 1];
    __shared__ double stage[LDU_MAX_PEERS][4];
    __shared__ double sh[4];          // lane 0's partial sums / broadcast scalars
    __shared__ int shStop;
    const int lane = threadIdx.x;
    const int nPF = C.nPF;
    for (int f = lane; f < nF; f += LDU_WAVE)
    {
        l[f] = gl[f]; u[f] = gu[f];
        upper[f] = gupper[f]; lower[f] = glower[f];
        if (BI) losort[f] = glosort[f];
    }
    for (int i = lane; i < n; i += LDU_WAVE)
    {
        const int o = perm[i];
        psi[o] = psiNew[i];
        b[o] = srcNew[i];
    }
    for (int c = lane; c < n; c += LDU_WAVE) diag[c] = gdiag[c];
    for (int i = lane; i < nPF; i += LDU_WAVE) { pf[i] = perm[C.pfCellNew[i]]; bou[i] = C.bou[i]; intc[i] = C.intc[i]; }
    unsigned hseq = C.kseq ? *C.kseq : 0u, rseq = *C.K.d_redSeq;
    __syncthreads();
    const double great_ = 1e20, small_ = 1e-20, vsmall_ = 1e-300;

    // wA = A psi (wT = T psi) with the interfaces, residuals, normFactor's local pieces
    co_halo(C, hseq, psi, pf, pnf);
    if (lane == 0)
    {
        for (int c = 0; c < n; c++) { wA[c] = diag[c] * psi[c]; if (BI) wT[c] = wA[c]; }
        for (int f = 0; f < nF; f++)
        {
            wA[u[f]] += lower[f] * psi[l[f]];
            wA[l[f]] += upper[f] * psi[u[f]];
            if (BI)
            {
                wT[u[f]] += upper[f] * psi[l[f]];
                wT[l[f]] += lower[f] * psi[u[f]];
            }
        }
        for (int i = 0; i < nPF; i++) { wA[pf[i]] -= bou[i] * pnf[i]; if (BI) wT[pf[i]] -= intc[i] * pnf[i]; }
        for (int c = 0; c < n; c++) { rA[c] = b[c] - wA[c]; if (BI) rT[c] = b[c] - wT[c]; }
        // sumA (lduMatrixATmul.C:167-202): interior, then the interfaces' boundary coefficients
        for (int c = 0; c < n; c++) pA[c] = diag[c];
        for (int f = 0; f < nF; f++) { pA[u[f]] += lower[f]; pA[l[f]] += upper[f]; }
        for (int i = 0; i < nPF; i++) pA[pf[i]] -= bou[i];
        double sum = 0.0;
        for (int c = 0; c < n; c++) sum += psi[c];
        sh[0] = sum; sh[1] = (double)n;
    }
    __syncthreads();
    double g = co_allreduce(C, rseq, lane < 2 ? sh[lane] : 0.0, 2, stage);   // gAverage(psi) = gSum / gSum(n)
    const double avg = __shfl(g, 0) / __shfl(g, 1);
    if (lane == 0)
    {
        for (int c = 0; c < n; c++) pA[c] *= avg;
        double nfl = 0.0, res = 0.0;
        for (int c = 0; c < n; c++) nfl += fabs(wA[c] - pA[c]) + fabs(b[c] - pA[c]);
        for (int c = 0; c < n; c++) res += fabs(rA[c]);
        sh[0] = nfl; sh[1] = res;
    }
    __syncthreads();
    g = co_allreduce(C, rseq, lane < 2 ? sh[lane] : 0.0, 2, stage);
    const double nf = __shfl(g, 0) + small_;
    const double initial = __shfl(g, 1) / nf;
    double final_ = initial;
    bool converged = final_ < tolerance || (relTol > small_ && final_ < relTol * initial);
    if (!converged)
    {
        if (lane == 0)
        {
            for (int c = 0; c < n; c++) rD[c] = diag[c];
            for (int f = 0; f < nF; f++) rD[u[f]] -= upper[f] * lower[f] / rD[l[f]];
            for (int c = 0; c < n; c++) rD[c] = 1.0 / rD[c];
            if (BI) for (int c = 0; c < n; c++) pT[c] = 0.0;
        }
        double wArA = great_, wArAold;
        int nIterations = 0;
        do
        {
            wArAold = wArA;
            if (lane == 0)
            {
                for (int c = 0; c < n; c++) wA[c] = rD[c] * rA[c];
                for (int f = 0; f < nF; f++)
                {
                    const int sf = BI ? losort[f] : f;
                    wA[u[sf]] -= rD[u[sf]] * lower[sf] * wA[l[sf]];
                }
                for (int f = nF - 1; f >= 0; f--) wA[l[f]] -= rD[l[f]] * upper[f] * wA[u[f]];
                if (BI)
                {
                    for (int c = 0; c < n; c++) wT[c] = rD[c] * rT[c];
                    for (int f = 0; f < nF; f++) wT[u[f]] -= rD[u[f]] * upper[f] * wT[l[f]];
                    for (int f = nF - 1; f >= 0; f--)
                    {
                        const int sf = losort[f];
                        wT[l[sf]] -= rD[l[sf]] * lower[sf] * wT[u[sf]];
                    }
                }
                double t = 0.0;
                for (int c = 0; c < n; c++) t += wA[c] * (BI ? rT[c] : rA[c]);
                sh[0] = t;
            }
            __syncthreads();
            g = co_allreduce(C, rseq, lane < 1 ? sh[0] : 0.0, 1, stage);
            wArA = __shfl(g, 0);
            if (lane == 0)
            {
                if (nIterations == 0)
                {
                    for (int c = 0; c < n; c++) { pA[c] = wA[c]; if (BI) pT[c] = wT[c]; }
                }
                else
                {
                    const double beta = wArA / wArAold;
                    for (int c = 0; c < n; c++)
                    {
                        pA[c] = wA[c] + beta * pA[c];
                        if (BI) pT[c] = wT[c] + beta * pT[c];
                    }
                }
            }
            __syncthreads();
            // wA = A pA: interfaces initialised, interior, interfaces updated (lduMatrixATmul.C:34-92)
            co_halo(C, hseq, pA, pf, pnf);
            if (lane == 0)
            {
                for (int c = 0; c < n; c++) wA[c] = diag[c] * pA[c];
                for (int f = 0; f < nF; f++)
                {
                    wA[u[f]] += lower[f] * pA[l[f]];
                    wA[l[f]] += upper[f] * pA[u[f]];
                }
                for (int i = 0; i < nPF; i++) wA[pf[i]] -= bou[i] * pnf[i];
            }
            if (BI)
            {
                __syncthreads();
                co_halo(C, hseq, pT, pf, pnf);     // Tmul (lduMatrixATmul.C:95-155): interfaceIntCoeffs
                if (lane == 0)
                {
                    for (int c = 0; c < n; c++) wT[c] = diag[c] * pT[c];
                    for (int f = 0; f < nF; f++)
                    {
                        wT[u[f]] += upper[f] * pT[l[f]];
                        wT[l[f]] += lower[f] * pT[u[f]];
                    }
                    for (int i = 0; i < nPF; i++) wT[pf[i]] -= intc[i] * pnf[i];
                }
            }
            if (lane == 0)
            {
                double t = 0.0;
                for (int c = 0; c < n; c++) t += wA[c] * (BI ? pT[c] : pA[c]);
                sh[0] = t;
            }
            __syncthreads();
            g = co_allreduce(C, rseq, lane < 1 ? sh[0] : 0.0, 1, stage);
            const double wApA = __shfl(g, 0);
            if (fabs(wApA) / nf < vsmall_) break;   // checkSingularity (uniform: every rank holds the same sum)
            const double alpha = wArA / wApA;
            if (lane == 0)
            {
                double res = 0.0;
                for (int c = 0; c < n; c++)
                {
                    psi[c] += alpha * pA[c];
                    rA[c] -= alpha * wA[c];
                    if (BI) rT[c] -= alpha * wT[c];
                }
                for (int c = 0; c < n; c++) res += fabs(rA[c]);
                sh[0] = res;
            }
            __syncthreads();
            // (second value: this rank's peer-timeout word - summed over the ranks, so that every rank leaves the loop in the
            //  SAME iteration when a wait gave up anywhere: the sequence numbers of the kernel-private exchanges stay equal)
            g = co_allreduce(C, rseq, lane < 1 ? sh[0] : (lane == 1 ? (double)C.abortFlag[LDU_PEER_FLAG] : 0.0), 2, stage);
            final_ = __shfl(g, 0) / nf;
            converged = final_ < tolerance || (relTol > small_ && final_ < relTol * initial);
            const double gStop = __shfl(g, 1);           // (every lane takes part: lane 1 is the source)
            if (lane == 0) shStop = gStop != 0.0 ? 1 : 0;
            __syncthreads();
            if (shStop) break;
        } while (nIterations++ < maxIter && !converged);
    }
    __syncthreads();
    if (lane == 0)
    {
        if (C.kseq) *C.kseq = hseq;
        *C.K.d_redSeq = rseq;
    }
    for (int i = lane; i < n; i += LDU_WAVE) psiNew[i] = psi[perm[i]];
}

int k_coarsest_set_peer_timeout(unsigned long long ticks)
{
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_peer_budget), &ticks, sizeof(ticks)));
    return 0;
}

// the distributed coarsest-level solve; 1 = not taken.  `eligible` = every rank can take it (decided once per hierarchy
// by the caller with an and-reduce: a kernel that communicates must be launched by all ranks or by none)
int k_coarsest_solve_peer(ldu_matrix* A, double tolerance, double relTol, int maxIter, double* corr, const double* src,
                          const int* d_cycPair)
{
    ldu_addr* a = A->a;
    ldu_ctx* ctx = a->ctx;
    CoarsePeer C;
    memset(&C, 0, sizeof(C));
    if (!comm_peer_kernel_comm(ctx, &C.K)) return 1;
    C.nPF = a->nPatchFaces;
    C.pfCellNew = a->d_pfCell;
    C.cycPair = d_cycPair;
    C.bou = A->d_bou;
    C.intc = A->d_int;
    C.abortFlag = ctx->d_abort;
    if (a->peer) { C.kdst = a->peer->d_kdst; C.ksrc = a->peer->d_ksrc; C.kseq = a->peer->d_kseq; }
    hipStream_t s = ctx->stream;
    if (A->sym)
        coarsest_krylov_peer_kernel<false><<<1, LDU_WAVE, 0, s>>>(C, a->nCells, a->nFaces, a->d_l, a->d_u, a->d_losort,
            A->d_diagO, A->d_upperO, A->d_upperO, a->d_perm, corr, src, tolerance, relTol, maxIter);
    else
        coarsest_krylov_peer_kernel<true><<<1, LDU_WAVE, 0, s>>>(C, a->nCells, a->nFaces, a->d_l, a->d_u, a->d_losort,
            A->d_diagO, A->d_upperO, A->d_lowerO, a->d_perm, corr, src, tolerance, relTol, maxIter);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}
// can THIS rank run its part of the distributed coarsest solve in the kernel above (sizes, peer regions)?
bool k_coarsest_peer_eligible(ldu_matrix* A)
{
    ldu_addr* a = A->a;
    static const bool off = getenv("LDU_COARSEST_KERNEL") && !atoi(getenv("LDU_COARSEST_KERNEL"));
    PeerKernelComm K;
    if (off || !comm_peer_kernel_comm(a->ctx, &K)) return false;
    if (a->nCells > CO_MAXC || a->nFaces > CO_MAXF || a->nPatchFaces > LDU_COARSEST_MAXP || a->nCells == 0) return false;
    bool remote = false;
    for (auto& P : a->patches) if (P.nbrPatch < 0 && P.n) remote = true;
    if (remote && !(a->peer && a->peer->kAll)) return false;
    return true;
}

// ---------------------------------------------------------------- directSolveCoarsest
// GAMGSolver.C:95-106 / GAMGSolverSolve.C:436-440: the coarsest level as a dense matrix (LUscalarMatrix.C:128-187
// convert), Crout LU with implicit scaled partial pivoting (scalarMatrices.C:31-134 LUDecompose) and
// LUBacksubstitute (scalarMatricesTemplates.C:119-164).  One wavefront, the matrix in LDS.  Decomposition: lane i owns
// row i; element (i, j) is  sum -= M[i][k] * M[k][j]  for k ascending (the reference's order) - M[k][j] is final once
// lane k has taken its k-1 earlier subtractions, so the k loop runs for all rows at once with lane k's sum broadcast
// at step k: the same products subtracted in the same order, n^2/2 steps instead of n^3/3.  Back-substitution is a
// chain from row to row (row i starts with the x just computed): one lane, the reference's loops literally.
// Serial systems without coupled patches, coarsest level <= 64 cells (nCellsInCoarsestLevel of the tutorials: 10-50).
#define LU_STRIDE (CO_MAXC + 1)
__global__ void __launch_bounds__(LDU_WAVE)
coarsest_lu_kernel(int n, int nF, const int* __restrict__ gl, const int* __restrict__ gu,
                   const double* __restrict__ gdiag, const double* __restrict__ gupper, const double* __restrict__ glower,
                   const int* __restrict__ perm, double* __restrict__ corrNew, const double* __restrict__ srcNew,
                   int* __restrict__ singular)
{
    __shared__ double M[CO_MAXC * LU_STRIDE];
    __shared__ double vv[CO_MAXC], x[CO_MAXC];
    __shared__ int piv[CO_MAXC];
    const int lane = threadIdx.x;
    for (int e = lane; e < CO_MAXC * LU_STRIDE; e += LDU_WAVE) M[e] = 0.0;
    __syncthreads();
    for (int c = lane; c < n; c += LDU_WAVE) M[c * LU_STRIDE + c] = gdiag[c];
    for (int f = lane; f < nF; f += LDU_WAVE)
    {
        M[gu[f] * LU_STRIDE + gl[f]] = glower[f];
        M[gl[f] * LU_STRIDE + gu[f]] = gupper[f];
    }
    for (int i = lane; i < n; i += LDU_WAVE) x[perm[i]] = srcNew[i];   // coarsestCorrField = coarsestSource (original numbering)
    __syncthreads();
    const bool row = lane < n;
    if (row)
    {
        double largest = 0.0;
        for (int j = 0; j < n; j++)
        {
            const double t = fabs(M[lane * LU_STRIDE + j]);
            if (t > largest) largest = t;
        }
        if (largest == 0.0) *singular = 1;   // the reference: FatalError "Singular matrix"
        vv[lane] = 1.0 / largest;
    }
    __syncthreads();
    for (int j = 0; j < n; j++)
    {
        double sum = row ? M[lane * LU_STRIDE + j] : 0.0;
        for (int k = 0; k < j; k++)
        {
            const double mkj = __shfl(sum, k);                 // M[k][j]: final after lane k's k-1 subtractions
            if (row && lane > k) sum -= M[lane * LU_STRIDE + k] * mkj;
        }
        if (row) M[lane * LU_STRIDE + j] = sum;
        // iMax: the LAST row i >= j with vv[i]*|sum| equal to the largest (the reference's `>=` scan)
        const double t = (row && lane >= j) ? vv[lane] * fabs(sum) : -1.0;
        double m = t;
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(m, off); m = o > m ? o : m; }
        if (!(m > 0.0)) m = 0.0;
        const unsigned long long hit = __ballot(row && lane >= j && t >= m);
        const int iMax = hit ? 63 - __builtin_clzll(hit) : j;
        __syncthreads();
        if (lane == 0) piv[j] = iMax;
        if (iMax != j)
        {
            if (row)
            {
                const double a = M[j * LU_STRIDE + lane], b = M[iMax * LU_STRIDE + lane];
                M[j * LU_STRIDE + lane] = b; M[iMax * LU_STRIDE + lane] = a;
            }
            __syncthreads();
            if (lane == 0) vv[iMax] = vv[j];
        }
        __syncthreads();
        if (lane == 0 && M[j * LU_STRIDE + j] == 0.0) M[j * LU_STRIDE + j] = 1e-15;   // SMALL
        __syncthreads();
        if (j != n - 1)
        {
            const double rDiag = 1.0 / M[j * LU_STRIDE + j];
            if (row && lane > j) M[lane * LU_STRIDE + j] *= rDiag;
        }
        __syncthreads();
    }
    if (lane == 0)
    {
        int ii = 0;
        for (int i = 0; i < n; i++)
        {
            const int ip = piv[i];
            double sum = x[ip];
            x[ip] = x[i];
            if (ii != 0) { for (int j = ii - 1; j < i; j++) sum -= M[i * LU_STRIDE + j] * x[j]; }
            else if (sum != 0.0) ii = i + 1;
            x[i] = sum;
        }
        for (int i = n - 1; i >= 0; i--)
        {
            double sum = x[i];
            for (int j = i + 1; j < n; j++) sum -= M[i * LU_STRIDE + j] * x[j];
            x[i] = sum / M[i * LU_STRIDE + i];
        }
    }
    __syncthreads();
    for (int i = lane; i < n; i += LDU_WAVE) corrNew[i] = x[perm[i]];
}

// ---------------------------------------------------------------- directSolveCoarsest with coupled patches / several ranks
// Serial run with cyclic patches: LUscalarMatrix.C:128-187 (the interfaces' coefficients subtracted from the dense matrix).
// Parallel run: the reference sends every rank's coarsest-level matrix to the master (LUscalarMatrix.C:52-107), builds ONE
// dense matrix over all ranks' cells there (:190-318), factorises it, and per V-cycle gathers the sources, back-substitutes
// and scatters (LUscalarMatrixTemplates.C:31-118).  Here the pieces are ALL-gathered (comm_allgather_host: the addressing
// once, the coefficients once per set of coefficients, the sources per V-cycle) and EVERY rank assembles, factorises and
// back-substitutes the same matrix - the master's arithmetic on every rank, the same bits, no scatter - and keeps its slice.
// Assembly in the reference's order: "set" entries (diag, lower / upper by face) are distinct and written in parallel; the
// coupling entries are subtracted after them by one lane in the reference's order (rank, patch, face).  One wavefront, the
// matrix in LDS, R = 1 or 2 rows per lane: up to 128 cells over all ranks (dynamic LDS 132 KB of the CU's 160).
#define LU_MAXN 128
struct CoarsestLU {
    int n = 0, myOff = 0, nMine = 0, nRanks = 0;
    int nSet = 0, nSub = 0;
    int* d_ops = nullptr;             // [3 * (nSet + nSub)] row, col, index into the gathered coefficients
    double* d_G = nullptr;            // gathered coefficients: per rank diag | upper | lower | bouCoeffs
    std::vector<int64_t> gOff;        // [nRanks + 1] offsets of the ranks' blocks in d_G (doubles)
    double* d_pack = nullptr;         // this rank's block
    double* d_M0 = nullptr;           // the assembled matrix, n x n
    double* d_srcAll = nullptr;       // [n] sources of all ranks, original numbering
    double* d_srcMine = nullptr;      // [nMine]
    std::vector<double> hA, hB;
    uint64_t epoch = ~0ull;
    int commEpoch = -1;
    std::string refuse;
    void release()
    {
        for (void* q : {(void*)d_ops, (void*)d_G, (void*)d_pack, (void*)d_M0, (void*)d_srcAll, (void*)d_srcMine})
            if (q) (void)hipFree(q);
        d_ops = nullptr; d_G = d_pack = d_M0 = d_srcAll = d_srcMine = nullptr;
    }
};
void coarsest_lu_free(ldu_matrix* m)
{
    if (!m->lu) return;
    m->lu->release();
    delete m->lu;
    m->lu = nullptr;
}

__global__ void __launch_bounds__(LDU_WAVE)
lu_assemble_kernel(int n, int nSet, int nSub, const int* __restrict__ ops, const double* __restrict__ G, double* __restrict__ M)
{
    const int lane = threadIdx.x;
    for (int e = lane; e < n * n; e += LDU_WAVE) M[e] = 0.0;
    __syncthreads();
    for (int o = lane; o < nSet; o += LDU_WAVE) M[ops[3 * o] * n + ops[3 * o + 1]] = G[ops[3 * o + 2]];
    __syncthreads();
    if (lane == 0)
        for (int o = nSet; o < nSet + nSub; o++) M[ops[3 * o] * n + ops[3 * o + 1]] -= G[ops[3 * o + 2]];
}

__global__ void lu_pack_src_kernel(int n, const int* __restrict__ perm, const double* __restrict__ srcNew, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[perm[i]] = srcNew[i];     // original numbering
}

// LUDecompose / LUBacksubstitute as coarsest_lu_kernel above, rows lane + 64 q (q < R) per lane
template <int R>
__global__ void __launch_bounds__(LDU_WAVE)
dense_lu_kernel(int n, const double* __restrict__ M0, const double* __restrict__ srcAll, int myOff, int nMine,
                const int* __restrict__ perm, double* __restrict__ corrNew, int* __restrict__ singular)
{
    extern __shared__ double lu_lds[];
    const int ld = n + 1;
    double* M = lu_lds;
    double* vv = M + n * ld;
    double* x = vv + n;
    int* piv = (int*)(x + n);
    const int lane = threadIdx.x;
    for (int e = lane; e < n * n; e += LDU_WAVE) { const int i = e / n; M[i * ld + (e - i * n)] = M0[e]; }
    for (int i = lane; i < n; i += LDU_WAVE) x[i] = srcAll[i];   // coarsestCorrField = coarsestSource
    __syncthreads();
    int ri[R];
    bool row[R];
#pragma unroll
    for (int q = 0; q < R; q++) { ri[q] = lane + LDU_WAVE * q; row[q] = ri[q] < n; }
#pragma unroll
    for (int q = 0; q < R; q++)
        if (row[q])
        {
            double largest = 0.0;
            for (int j = 0; j < n; j++)
            {
                const double t = fabs(M[ri[q] * ld + j]);
                if (t > largest) largest = t;
            }
            if (largest == 0.0) *singular = 1;   // the reference: FatalError "Singular matrix"
            vv[ri[q]] = 1.0 / largest;
        }
    __syncthreads();
    for (int j = 0; j < n; j++)
    {
        double sum[R];
#pragma unroll
        for (int q = 0; q < R; q++) sum[q] = row[q] ? M[ri[q] * ld + j] : 0.0;
        for (int k = 0; k < j; k++)
        {
            double mkj;                                       // M[k][j]: final after row k's k-1 subtractions
            if (R == 1) mkj = __shfl(sum[0], k);
            else mkj = k < LDU_WAVE ? __shfl(sum[0], k) : __shfl(sum[R - 1], k - LDU_WAVE);
#pragma unroll
            for (int q = 0; q < R; q++)
                if (row[q] && ri[q] > k) sum[q] -= M[ri[q] * ld + k] * mkj;
        }
        double m = -1.0, t[R];
#pragma unroll
        for (int q = 0; q < R; q++)
        {
            if (row[q]) M[ri[q] * ld + j] = sum[q];
            t[q] = (row[q] && ri[q] >= j) ? vv[ri[q]] * fabs(sum[q]) : -1.0;
            m = t[q] > m ? t[q] : m;
        }
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(m, off); m = o > m ? o : m; }
        if (!(m > 0.0)) m = 0.0;
        int iMax = j;                                          // the LAST row i >= j whose vv[i]*|sum| equals the largest (`>=` scan)
#pragma unroll
        for (int q = 0; q < R; q++)
        {
            const unsigned long long hit = __ballot(row[q] && ri[q] >= j && t[q] >= m);
            if (hit) iMax = LDU_WAVE * q + 63 - __builtin_clzll(hit);
        }
        __syncthreads();
        if (lane == 0) piv[j] = iMax;
        if (iMax != j)
        {
#pragma unroll
            for (int q = 0; q < R; q++)
                if (row[q])
                {
                    const double a = M[j * ld + ri[q]], b = M[iMax * ld + ri[q]];
                    M[j * ld + ri[q]] = b; M[iMax * ld + ri[q]] = a;
                }
            __syncthreads();
            if (lane == 0) vv[iMax] = vv[j];
        }
        __syncthreads();
        if (lane == 0 && M[j * ld + j] == 0.0) M[j * ld + j] = 1e-15;   // SMALL
        __syncthreads();
        if (j != n - 1)
        {
            const double rDiag = 1.0 / M[j * ld + j];
#pragma unroll
            for (int q = 0; q < R; q++)
                if (row[q] && ri[q] > j) M[ri[q] * ld + j] *= rDiag;
        }
        __syncthreads();
    }
    if (lane == 0)
    {
        int ii = 0;
        for (int i = 0; i < n; i++)
        {
            const int ip = piv[i];
            double sum = x[ip];
            x[ip] = x[i];
            if (ii != 0) { for (int j = ii - 1; j < i; j++) sum -= M[i * ld + j] * x[j]; }
            else if (sum != 0.0) ii = i + 1;
            x[i] = sum;
        }
        for (int i = n - 1; i >= 0; i--)
        {
            double sum = x[i];
            for (int j = i + 1; j < n; j++) sum -= M[i * ld + j] * x[j];
            x[i] = sum / M[i * ld + i];
        }
    }
    __syncthreads();
    for (int i = lane; i < nMine; i += LDU_WAVE) corrNew[i] = x[myOff + perm[i]];
}

// the gathered addressing -> the assembly table (collective: every rank calls it at the same point of the same V-cycle)
static int lu_gathered_setup(ldu_matrix* A, CoarsestLU* L)
{
    ldu_addr* a = A->a;
    ldu_ctx* ctx = a->ctx;
    L->release();
    L->refuse.clear();
    L->commEpoch = ctx->commEpoch;
    L->epoch = ~0ull;
    // this rank's description: nCells, nFaces, nPatches, l, u, then per patch: nbrRank, nbrPatch, n, offset, faceCells
    std::vector<int> d = {a->nCells, a->nFaces, (int)a->patches.size()};
    d.insert(d.end(), a->l.begin(), a->l.end());
    d.insert(d.end(), a->u.begin(), a->u.end());
    for (auto& P : a->patches)
    {
        d.push_back(P.nbrRank); d.push_back(P.nbrPatch); d.push_back(P.n); d.push_back(P.offset);
        d.insert(d.end(), P.faceCells.begin(), P.faceCells.end());
    }
    std::vector<std::vector<char>> all;
    if (comm_allgather_host(ctx, d.data(), (int64_t)(sizeof(int) * d.size()), all)) return -1;
    const int nR = (int)all.size(), me = nR > 1 ? ctx->rank : 0;
    struct RP { int nbrRank, nbrPatch, n, offset; const int* fc; };
    struct RD { int nC, nF; const int *l, *u; std::vector<RP> P; int nPF; };
    std::vector<RD> D(nR);
    std::vector<int> cellOff(nR + 1, 0);
    L->gOff.assign(nR + 1, 0);
    for (int r = 0; r < nR; r++)
    {
        const int* q = (const int*)all[r].data();
        RD& X = D[r];
        X.nC = q[0]; X.nF = q[1];
        const int nP = q[2];
        X.l = q + 3; X.u = X.l + X.nF;
        const int* w = X.u + X.nF;
        X.nPF = 0;
        for (int p = 0; p < nP; p++)
        {
            X.P.push_back(RP{w[0], w[1], w[2], w[3], w + 4});
            X.nPF = std::max(X.nPF, w[3] + w[2]);
            w += 4 + w[2];
        }
        cellOff[r + 1] = cellOff[r] + X.nC;
        L->gOff[r + 1] = L->gOff[r] + X.nC + 2 * (int64_t)X.nF + X.nPF;
    }
    L->nRanks = nR;
    L->n = cellOff[nR];
    L->myOff = cellOff[me];
    L->nMine = a->nCells;
    // (every rank sees the same description: the refusals below are taken by all of them)
    if (L->n > LU_MAXN || L->n == 0)
    {
        L->refuse = "directSolveCoarsest: the coarsest level has " + std::to_string(L->n) + " cells over all ranks; the device LU holds up to "
                    + std::to_string(LU_MAXN);
        return 0;
    }
    std::vector<int> set, sub;
    auto op = [](std::vector<int>& v, int row, int col, int64_t src) { v.push_back(row); v.push_back(col); v.push_back((int)src); };
    for (int r = 0; r < nR; r++)
    {
        const RD& X = D[r];
        const int off = cellOff[r];
        const int64_t g = L->gOff[r], gU = g + X.nC, gL = gU + X.nF, gB = gL + X.nF;
        for (int c = 0; c < X.nC; c++) op(set, off + c, off + c, g + c);
        for (int f = 0; f < X.nF; f++)
        {
            op(set, off + X.u[f], off + X.l[f], gL + f);
            op(set, off + X.l[f], off + X.u[f], gU + f);
        }
        for (int p = 0; p < (int)X.P.size(); p++)
        {
            const RP& P = X.P[p];
            if (P.nbrPatch >= 0)
            {
                if (nR > 1)
                {
                    // procLduInterface.C:38-60 gives a cyclic interface myProcNo == neighbProcNo == -1, and
                    // LUscalarMatrix.C:247-262 then reads it as ONE patch holding both halves: no 2.2.x mesh has that layout
                    L->refuse = "directSolveCoarsest: a cyclic patch inside a rank of a parallel run (the reference reads it as the "
                                "pre-2.0 single-patch cyclic, LUscalarMatrix.C:247-262) is not implemented";
                    return 0;
                }
                const RP& N = X.P[P.nbrPatch];                 // LUscalarMatrix.C:160-184
                for (int f = 0; f < P.n; f++) op(sub, P.fc[f], N.fc[f], gB + N.offset + f);
                continue;
            }
            if (nR == 1 || P.nbrRank < 0 || P.nbrRank >= nR || P.nbrRank == r)
            {
                L->refuse = "directSolveCoarsest: a processor patch without a neighbour rank in this communicator";
                return 0;
            }
            if (r > P.nbrRank) continue;                        // :264 myProcNo_ < neighbProcNo_
            // the neighbour's interface: its k-th patch towards r for my k-th patch towards it (the pairing of every exchange,
            // ldu_comm.cpp paired_patch; the reference matches by the communication tag, :279-292)
            int k = 0, j = -1;
            for (int i = 0; i < p; i++) if (X.P[i].nbrPatch < 0 && X.P[i].nbrRank == P.nbrRank) k++;
            const RD& Y = D[P.nbrRank];
            for (int i = 0; i < (int)Y.P.size(); i++)
                if (Y.P[i].nbrPatch < 0 && Y.P[i].nbrRank == r && k-- == 0) { j = i; break; }
            if (j < 0 || Y.P[j].n != P.n) { L->refuse = "directSolveCoarsest: unpaired processor patch"; return 0; }
            const RP& N = Y.P[j];
            const int noff = cellOff[P.nbrRank];
            const int64_t gBN = L->gOff[P.nbrRank] + Y.nC + 2 * (int64_t)Y.nF;
            for (int f = 0; f < P.n; f++)                        // :308-315
            {
                const int uCell = P.fc[f] + off, lCell = N.fc[f] + noff;
                op(sub, uCell, lCell, gBN + N.offset + f);
                op(sub, lCell, uCell, gB + P.offset + f);
            }
        }
    }
    L->nSet = (int)set.size() / 3;
    L->nSub = (int)sub.size() / 3;
    set.insert(set.end(), sub.begin(), sub.end());
    LDU_CHECK_HIP(hipMalloc((void**)&L->d_ops, sizeof(int) * (set.size() + 3)));
    LDU_CHECK_HIP(hipMemcpy(L->d_ops, set.data(), sizeof(int) * set.size(), hipMemcpyHostToDevice));
    LDU_CHECK_HIP(hipMalloc((void**)&L->d_G, sizeof(double) * (size_t)(L->gOff[nR] + 1)));
    LDU_CHECK_HIP(hipMalloc((void**)&L->d_pack, sizeof(double) * (size_t)(L->gOff[me + 1] - L->gOff[me] + 1)));
    LDU_CHECK_HIP(hipMalloc((void**)&L->d_M0, sizeof(double) * (size_t)L->n * L->n));
    LDU_CHECK_HIP(hipMalloc((void**)&L->d_srcAll, sizeof(double) * (size_t)L->n));
    LDU_CHECK_HIP(hipMalloc((void**)&L->d_srcMine, sizeof(double) * (size_t)(L->nMine + 1)));
    // (per device: every set-up sets it for the device this context runs on)
    LDU_CHECK_HIP(hipFuncSetAttribute((const void*)dense_lu_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)((LU_MAXN * (LU_MAXN + 1) + 2 * LU_MAXN) * sizeof(double) + LU_MAXN * sizeof(int))));
    if (getenv("LDU_VERBOSE"))
        fprintf(stderr, "[ldugpu] directSolveCoarsest: %d cells over %d rank(s), %d matrix entries set, %d coupling entries; every rank "
                        "factorises the gathered matrix (one wavefront, %d row(s) per lane)\n", L->n, nR, L->nSet, L->nSub, L->n > LDU_WAVE ? 2 : 1);
    return 0;
}

static int lu_gathered(ldu_matrix* A, double* corr, const double* src, uint64_t epoch)
{
    ldu_addr* a = A->a;
    ldu_ctx* ctx = a->ctx;
    hipStream_t s = ctx->stream;
    if (!A->lu) A->lu = new CoarsestLU();
    CoarsestLU* L = A->lu;
    if (L->commEpoch != ctx->commEpoch || L->nMine != a->nCells)
        if (lu_gathered_setup(A, L)) return -1;
    if (!L->refuse.empty()) { ldu_set_error(L->refuse); return -1; }
    const int me = L->nRanks > 1 ? ctx->rank : 0;
    const size_t nC = (size_t)a->nCells, nF = (size_t)a->nFaces, nPF = (size_t)(L->gOff[me + 1] - L->gOff[me]) - nC - 2 * nF;
    if (L->epoch != epoch)
    {
        // this rank's coefficients (original numbering: the LDU arrays as they are), then everybody's
        double* dst = L->nRanks > 1 ? L->d_pack : L->d_G;
        if (nC) LDU_CHECK_HIP(hipMemcpyAsync(dst, A->d_diagO, sizeof(double) * nC, hipMemcpyDeviceToDevice, s));
        if (nF) LDU_CHECK_HIP(hipMemcpyAsync(dst + nC, A->d_upperO, sizeof(double) * nF, hipMemcpyDeviceToDevice, s));
        if (nF) LDU_CHECK_HIP(hipMemcpyAsync(dst + nC + nF, A->sym ? A->d_upperO : A->d_lowerO, sizeof(double) * nF, hipMemcpyDeviceToDevice, s));
        if (nPF) LDU_CHECK_HIP(hipMemcpyAsync(dst + nC + 2 * nF, A->d_bou, sizeof(double) * nPF, hipMemcpyDeviceToDevice, s));
        if (L->nRanks > 1)
        {
            L->hA.resize(nC + 2 * nF + nPF + 1);
            LDU_CHECK_HIP(hipMemcpyAsync(L->hA.data(), L->d_pack, sizeof(double) * (nC + 2 * nF + nPF), hipMemcpyDeviceToHost, s));
            LDU_CHECK_HIP(hipStreamSynchronize(s));
            std::vector<std::vector<char>> all;
            if (comm_allgather_host(ctx, L->hA.data(), (int64_t)(sizeof(double) * (nC + 2 * nF + nPF)), all)) return -1;
            L->hB.resize((size_t)L->gOff[L->nRanks] + 1);
            for (int r = 0; r < L->nRanks; r++)
            {
                if ((int64_t)all[r].size() != (int64_t)sizeof(double) * (L->gOff[r + 1] - L->gOff[r]))
                {
                    ldu_set_error("directSolveCoarsest: a rank's coefficient block does not match its addressing");
                    return -1;
                }
                memcpy(L->hB.data() + L->gOff[r], all[r].data(), all[r].size());
            }
            LDU_CHECK_HIP(hipMemcpyAsync(L->d_G, L->hB.data(), sizeof(double) * (size_t)L->gOff[L->nRanks], hipMemcpyHostToDevice, s));
        }
        lu_assemble_kernel<<<1, LDU_WAVE, 0, s>>>(L->n, L->nSet, L->nSub, L->d_ops, L->d_G, L->d_M0);
        LDU_CHECK_HIP(hipGetLastError());
        if (L->nRanks > 1) LDU_CHECK_HIP(hipStreamSynchronize(s));     // (hB is reused)
        L->epoch = epoch;
    }
    // the sources of all ranks, original numbering
    if (nC)
    {
        lu_pack_src_kernel<<<((int)nC + 63) / 64, 64, 0, s>>>((int)nC, a->d_perm, src, L->nRanks > 1 ? L->d_srcMine : L->d_srcAll);
        LDU_CHECK_HIP(hipGetLastError());
    }
    if (L->nRanks > 1)
    {
        L->hA.resize(nC + 1);
        if (nC) LDU_CHECK_HIP(hipMemcpyAsync(L->hA.data(), L->d_srcMine, sizeof(double) * nC, hipMemcpyDeviceToHost, s));
        LDU_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<std::vector<char>> all;
        if (comm_allgather_host(ctx, L->hA.data(), (int64_t)(sizeof(double) * nC), all)) return -1;
        L->hB.resize((size_t)L->n + 1);
        size_t off = 0;
        for (int r = 0; r < L->nRanks; r++) { memcpy(L->hB.data() + off, all[r].data(), all[r].size()); off += all[r].size() / sizeof(double); }
        if ((int)off != L->n) { ldu_set_error("directSolveCoarsest: the gathered sources do not match the gathered addressing"); return -1; }
        LDU_CHECK_HIP(hipMemcpyAsync(L->d_srcAll, L->hB.data(), sizeof(double) * (size_t)L->n, hipMemcpyHostToDevice, s));
    }
    const size_t lds = ((size_t)L->n * (L->n + 1) + 2 * (size_t)L->n) * sizeof(double) + (size_t)L->n * sizeof(int);
    if (L->n <= LDU_WAVE)
        dense_lu_kernel<1><<<1, LDU_WAVE, lds, s>>>(L->n, L->d_M0, L->d_srcAll, L->myOff, L->nMine, a->d_perm, corr, ctx->d_abort + 1);
    else
        dense_lu_kernel<2><<<1, LDU_WAVE, lds, s>>>(L->n, L->d_M0, L->d_srcAll, L->myOff, L->nMine, a->d_perm, corr, ctx->d_abort + 1);
    LDU_CHECK_HIP(hipGetLastError());
    if (L->nRanks > 1) LDU_CHECK_HIP(hipStreamSynchronize(s));         // (hB is reused by the next call)
    return 0;
}

// 0 = solved; -1 = error (said); directSolveCoarsest has no iterative fall-back
int k_coarsest_lu(ldu_matrix* A, double* corr, const double* src, uint64_t epoch)
{
    ldu_addr* a = A->a;
    ldu_ctx* ctx = a->ctx;
    if ((ctx->comm && ctx->nRanks > 1) || a->nPatchFaces || !a->patches.empty()) return lu_gathered(A, corr, src, epoch);
    if (a->nCells > CO_MAXC || a->nCells == 0)
    {
        ldu_set_error("directSolveCoarsest: the coarsest level has " + std::to_string(a->nCells) + " cells; the device LU holds up to "
                      + std::to_string(CO_MAXC));
        return -1;
    }
    hipStream_t s = ctx->stream;
    coarsest_lu_kernel<<<1, LDU_WAVE, 0, s>>>(a->nCells, a->nFaces, a->d_l, a->d_u, A->d_diagO, A->d_upperO,
        A->sym ? A->d_upperO : A->d_lowerO, a->d_perm, corr, src, ctx->d_abort + 1);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}
