// Host-side solver loops driving the HIP kernels: PCG (PCG.C:65-182), PBiCG (PBiCG.C:65-198),
// smoothSolver (smoothSolver.C:77-180), diagonalSolver, the preconditioners and smoothers.
// All vectors are device arrays in the level-ordered numbering.  Scalars (wArA, wApA, ...)
// stay on the device; the host reads back only the residual it needs for checkConvergence.
#include <chrono>
#include <cmath>
#include <cstring>

#include "ldu_internal.hpp"

static const double kGreat = 1e20;    // SolverPerformance.H:269-275
static const double kSmall = 1e-20;

double* ldu_matrix::workVec(int i)
{
    while ((int)work.size() <= i) work.push_back(nullptr);
    if (!work[i])
    {
        size_t n = (size_t)(a->nCells > 0 ? a->nCells : 1) + 64;
        if (hipMalloc((void**)&work[i], n * sizeof(double)) != hipSuccess) return nullptr;
    }
    return work[i];
}

// The four flag words behind the scalars (d_abort / h_abort): [0] a point-to-point sweep gave up a dependency wait (->
// collective engine fallback), [1] singular coarsest-level matrix (directSolveCoarsest), [2] a wait for ANOTHER RANK timed
// out (peer-store backend, LDU_PEER_FLAG in ldu_peer_dev.hpp; fatal for the operation, no re-run), [3] spare.
static int abort_words_to_error(ldu_ctx* ctx, const int* h, hipStream_t s)
{
    if (h[2])
    {
        // the neighbour never wrote what this rank waited for: nothing a re-run on this rank alone could repair
        (void)hipStreamSynchronize(s);
        (void)hipMemsetAsync(ctx->d_abort, 0, 4 * sizeof(int), s);
        ctx->p2pGen++;
        ldu_set_error("peer-store backend: a wait for another rank timed out (LDU_PEER_TIMEOUT_S)");
        return -21;
    }
    if (h[1])
    {
        // LUDecompose's FatalError("Singular matrix") (scalarMatrices.C:52-56): a zero row of the coarsest level
        (void)hipMemsetAsync(ctx->d_abort, 0, 2 * sizeof(int), s);     // (and a sweep abort of the same operation with it)
        if (h[0]) ctx->p2pGen++;
        ldu_set_error("directSolveCoarsest: singular coarsest-level matrix");
        return -17;
    }
    if (h[0])
    {
        // a point-to-point sweep gave up waiting (bounded spin): fail loudly, never hang
        (void)hipMemsetAsync(ctx->d_abort, 0, sizeof(int), s);
        ctx->p2pGen++;
        ctx->abortSeen = 1;
        ldu_set_error("point-to-point sweep aborted: dependency wait exceeded its spin bound");
        return -20;
    }
    return 0;
}

int dev_read_scalars(ldu_ctx* ctx, int slot, int count, double* out)
{
    if (comm_allreduce_abort(ctx, ctx->stream)) return -1;
    ctx->nScalarReadbacks++;
    // scalars and the flag words (stored behind them) in one copy: a second small copy costs ~20 us of latency
    LDU_CHECK_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, sizeof(double) * (S_NSLOTS + 2), hipMemcpyDeviceToHost,
                                 ctx->stream));
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < count; i++) out[i] = ctx->h_scalars[ctx->sb + slot + i];
    const int rc = abort_words_to_error(ctx, ctx->h_abort, ctx->stream);
    ctx->h_abort[0] = ctx->h_abort[1] = ctx->h_abort[2] = 0;
    return rc;
}

// Surface an aborted point-to-point sweep to callers that do not read scalars (ldu_smooth, ...).
int dev_check_abort(ldu_ctx* ctx)
{
    if (comm_allreduce_abort(ctx, ctx->stream)) return -1;
    LDU_CHECK_HIP(hipMemcpyAsync(ctx->h_abort, ctx->d_abort, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const int rc = abort_words_to_error(ctx, ctx->h_abort, ctx->stream);
    ctx->h_abort[0] = ctx->h_abort[1] = ctx->h_abort[2] = 0;
    return rc;
}

int fallback_prepare(ldu_matrix* m)
{
    ldu_ctx* ctx = m->a->ctx;
    // whatever the failed attempt left in flight (the second stream of PBiCG included) has to end first
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream2));
    if (ctx->streamComm) LDU_CHECK_HIP(hipStreamSynchronize(ctx->streamComm));   // a halo exchange of the failed attempt
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->haloInFlight = false;
    LDU_CHECK_HIP(ldu_memset_sync(ctx->d_abort, 0, 4 * sizeof(int)));   // the abort flag and the words behind it
    ctx->h_abort[0] = ctx->h_abort[1] = ctx->h_abort[2] = 0;
    ctx->dualActive = 0;
    ctx->sb = 0;
    ctx->abortSeen = 0;
    // factors a broken sweep may have written (calcReciprocalD runs as a sweep): this matrix, its GAMG levels,
    // the coupled family's rD
    m->rDKind = -1;
    if (m->a->peer) m->a->peer->pending = false;      // a halo exchange of the failed attempt that was packed, never unpacked
    if (m->gamg) gamg_invalidate_factors(m->gamg);
    coupled_invalidate(m);
    if (ctx->nFallbacks++ == 0 || getenv("LDU_VERBOSE"))
        fprintf(stderr, "[ldugpu] warning: a point-to-point sweep exceeded its dependency-wait bound; the operation is "
                        "re-run on the level-kernel engine (fallback %ld of this context)\n", ctx->nFallbacks);
    return 0;
}

static int dev_write_scalar(ldu_ctx* ctx, int slot, double v)
{
    // stream-ordered write through a tiny staging slot in pinned memory is racy if reused
    // before the copy ran; use a blocking small copy instead (rare: once per solve).
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    LDU_CHECK_HIP(hipMemcpy(ctx->S() + slot, &v, sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

// ---------------------------------------------------------------- interface-aware matrix ops
// initMatrixInterfaces -> pack + exchange ; interior rows ; updateMatrixInterfaces -> apply
// (lduMatrixUpdateMatrixInterfaces.C:30-266)

int dev_halo_start(ldu_matrix* m, const double* x)
{
    ldu_addr* a = m->a;
    if (!a->nPatchFaces) return 0;
    return comm_halo_pack_exchange(a, x, a->ctx->stream);
}

int dev_amul(ldu_matrix* m, double* y, const double* x, bool transpose, hipStream_t s2)
{
    hipStream_t s = s2 ? s2 : m->a->ctx->stream;
    if (dev_halo_start(m, x)) return -1;
    if (k_amul(m, y, x, transpose, s)) return -1;
    if (m->a->nPatchFaces) return k_apply_patches(m->a, y, transpose ? m->d_int : m->d_bou, 1.0, s);
    return 0;
}

int dev_residual(ldu_matrix* m, double* r, const double* x, const double* b)
{
    hipStream_t s = m->a->ctx->stream;
    if (dev_halo_start(m, x)) return -1;
    if (k_residual_rows(m, r, x, b, s)) return -1;
    if (m->a->nPatchFaces) return k_apply_patches(m->a, r, m->d_bou, -1.0, s);
    return 0;
}

int dev_sumA(ldu_matrix* m, double* sumA)
{
    hipStream_t s = m->a->ctx->stream;
    if (k_sumA_rows(m, sumA, s)) return -1;
    if (m->a->nPatchFaces) return k_sumA_patches(m->a, sumA, m->d_bou, s);
    return 0;
}

// ---------------------------------------------------------------- preconditioners

int matrix_ensure_rD(ldu_matrix* m, int kind)
{
    ldu_addr* a = m->a;
    hipStream_t s = a->ctx->stream;
    if (kind == LDU_PRE_FDIC) kind = LDU_PRE_DIC;   // FDIC == DIC with precomputed rD*upper
    if (kind == LDU_PRE_DIAGONAL)
    {
        if (m->rDiagValid) return 0;
        if (!m->d_rDiag) LDU_CHECK_HIP(hipMalloc((void**)&m->d_rDiag, sizeof(double) * (size_t)(a->nCells + 1)));
        if (k_ew(a->nCells, EW_COPY, m->d_rDiag, m->d_diag, nullptr, s)) return -1;
        if (k_reciprocal(a->nCells, m->d_rDiag, s)) return -1;   // diagonalPreconditioner.C:62-68
        m->rDiagValid = true;
        return 0;
    }
    if (m->rDKind == kind) return 0;
    const size_t nE = (size_t)(a->nEntries > 0 ? a->nEntries : 1);
    if (!m->d_rD) LDU_CHECK_HIP(hipMalloc((void**)&m->d_rD, sizeof(double) * (size_t)(a->nCells + 1)));
    if (!m->d_valP) LDU_CHECK_HIP(hipMalloc((void**)&m->d_valP, sizeof(double) * nE));
    // calcReciprocalD (DICPreconditioner.C:57-84 / DILUPreconditioner.C:57-85)
    SweepArgs g{};
    g.mode = SW_RD;
    g.w = m->d_rD;
    g.scale = m->d_diag;
    g.val = m->d_valA;
    g.val2 = m->d_valT;
    if (k_sweep(a, g)) return -1;
    if (k_reciprocal(a->nCells, m->d_rD, s)) return -1;
    // rD[row]*coeff, the products the sweeps use (identical rounding: FDICPreconditioner.C:79-83)
    if (k_scale_rows(a, m->d_valP, m->d_valA, m->d_rD, s)) return -1;
    if (!m->sym)
    {
        if (!m->d_valPT) LDU_CHECK_HIP(hipMalloc((void**)&m->d_valPT, sizeof(double) * nE));
        if (k_scale_rows(a, m->d_valPT, m->d_valT, m->d_rD, s)) return -1;
    }
    m->rDKind = kind;
    return 0;
}

int dev_precondition(ldu_matrix* m, int kind, double* w, const double* r, bool transpose, hipStream_t s)
{
    ldu_addr* a = m->a;
    if (!s) s = a->ctx->stream;
    const int lane = (s == a->ctx->stream2) ? 1 : 0;   // second concurrent sweep: own P2P state
    switch (kind)
    {
    case LDU_PRE_NONE:       // noPreconditioner.C:58-74
        return k_ew(a->nCells, EW_COPY, w, r, nullptr, s);
    case LDU_PRE_DIAGONAL:   // diagonalPreconditioner.C:72-87
        if (matrix_ensure_rD(m, kind)) return -1;
        return k_ew(a->nCells, EW_MUL, w, m->d_rDiag, r, s);
    case LDU_PRE_DIC:
    case LDU_PRE_FDIC:
    case LDU_PRE_DILU:
    {
        if (matrix_ensure_rD(m, kind)) return -1;
        // precondition : forward uses lower[] on the neighbour rows, backward upper[] on owned rows
        // preconditionT: forward uses upper[], backward lower[]  (DILUPreconditioner.C:138-185)
        const double* vp = (transpose && !m->sym) ? m->d_valPT : m->d_valP;
        SweepArgs f{};
        f.mode = SW_TRI_FWD; f.w = w; f.rhs = r; f.scale = m->d_rD; f.val = vp;
        f.lane = lane; f.stream = s;
        if (k_sweep(a, f)) return -1;
        SweepArgs b{};
        b.mode = SW_TRI_BWD; b.w = w; b.val = vp;
        b.lane = lane; b.stream = s;
        return k_sweep(a, b);
    }
    }
    ldu_set_error("unknown preconditioner");
    return -3;
}

// ---------------------------------------------------------------- smoothers

int k_sweep_gs_blocks_if(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val, const double* bou);   // ldu_blocks.hip

static int smooth_gs(ldu_matrix* m, double* psi, const double* source, int nSweeps, bool sym)
{
    ldu_addr* a = m->a;
    hipStream_t s = a->ctx->stream;
    double* bPrime = nullptr;
    if (a->nPatchFaces || sym) bPrime = m->workVec(12);
    // LDS-resident blocks (ldu_blocks.hip) wherever the cluster engine (structured numberings, <= 6 + 6 neighbours) does not apply
    // (pend: this level's sweep plans are still being built on a host thread - ensure_hierarchy -: one sweep per launch on the
    //  level engines until they are there; the results are the same bit for bit)
    const bool pend = addr_bg_pending(a);
    {
        // (stand-alone smoothing calls on an addressing with processor patches - tests, tools -: the collective decision about
        //  the block engine is taken here when the caller says that every rank makes this call, LDU_BLK_PEER_FORCE=1; inside a
        //  GAMG solve it was taken for all levels at once, gamg_decide_peer_smoothers)
        if (!sym && a->ctx->comm && a->ctx->nRanks > 0 && a->peerBlkEpoch != a->ctx->commEpoch)
        {
            // (every rank, with or without coupled faces on this addressing: the decision is an all-reduce)
            const char* fe = getenv("LDU_BLK_PEER_FORCE");      // (read per call: tests switch it within one process)
            if (fe && atoi(fe) && k_blocks_peer_decide(a)) return -1;
        }
    }
    const bool blk = !pend && !sym && !a->nPatchFaces && a->ctx->sweepP2P && !(a->ctx->clusterMulti && k_cluster_active(a)) && k_blocks_active(a);
    if (!pend && !sym && !a->nPatchFaces && a->ctx->sweepP2P && !blk)
    {
        // small matrix: every sweep inside one workgroup, solution vector in LDS
        int rc = 1;
        if (nSweeps <= 4) rc = k_sweep_gs_wg(a, nSweeps, psi, source, m->d_diag, m->d_valA);
        else
        {
            // (more than four sweeps: launches of 4 / 3 / 2, never a single sweep left over)
            int left = nSweeps;
            rc = 0;
            while (left > 0 && rc == 0)
            {
                const int kk = left > 4 ? (left == 5 ? 3 : 4) : left;
                rc = k_sweep_gs_wg(a, kk, psi, source, m->d_diag, m->d_valA);
                if (rc > 0 && left != nSweeps) { ldu_set_error("workgroup engine: refused after the first launch"); return -1; }
                left -= kk;
            }
        }
        if (rc > 0) rc = k_sweep_gs_small(a, nSweeps, psi, source, m->d_diag, m->d_valA);
        if (rc <= 0) return rc;
    }
    if (blk && nSweeps == 1)
    {
        const int rc = k_sweep_gs_blocks(a, 1, psi, source, m->d_diag, m->d_valA);   // (LDS-resident blocks: ldu_blocks.hip)
        if (rc <= 0) return rc;
    }
    if (!pend && !sym && !a->nPatchFaces && a->ctx->sweepP2P && a->ctx->gsPipeline && nSweeps >= 2)
    {
        // consecutive sweeps pipelined inside one launch (bit-identical to separate sweeps)
        int left = nSweeps;
        bool pipelined = true;
        while (left > 0)
        {
            const int k = left > 4 ? 4 : left;
            // LDS-resident blocks (ldu_blocks.hip), clusters, then the level engines
            int rc = blk ? k_sweep_gs_blocks(a, k, psi, source, m->d_diag, m->d_valA) : 1;
            if (k == 1 && rc > 0) break;   // a single remaining sweep: the plain engine below
            if (rc > 0) rc = a->ctx->clusterMulti ? k_sweep_cluster_gs_multi(a, k, psi, source, m->d_diag, m->d_valA) : 1;
            if (rc > 0) rc = k_sweep_gs_multi(a, k, psi, source, m->d_diag, m->d_valA);
            if (rc < 0) return -1;
            if (rc > 0) { pipelined = false; break; }   // DAG too skewed: sweep by sweep
            left -= k;
        }
        if (pipelined && left == 0) return 0;
        nSweeps = left;
    }
    if (a->nPatchFaces && !sym && a->ctx->sweepP2P && nSweeps > 0 && k_blocks_active(a))
    {
        // coupled patches that are all cyclic (sub-domain mode: K ranks of the reference inside this addressing): the sweeps
        // pipelined on the block engine, the interface terms as entries of the rows (ldu_blocks.hip)
        int left = nSweeps;
        while (left > 0)
        {
            const int k = left > 4 ? 4 : left;
            const int rc = k_sweep_gs_blocks_if(a, k, psi, source, m->d_diag, m->d_valA, m->d_bou);
            if (rc < 0) return -1;
            if (rc > 0)
            {
                if (left != nSweeps) { ldu_set_error("block engine: refused after the first launch"); return -1; }
                break;
            }
            left -= k;
        }
        if (left == 0) return 0;
    }
    if (a->nPatchFaces && !sym && a->ctx->sweepP2P && a->peerWg == 1 && a->peerWgEpoch == a->ctx->commEpoch && nSweeps > 0)
    {
        // small GAMG level with coupled patches, peer-store backend: all sweeps of this smoothing AND their boundary
        // exchanges in one launch (collective decision: gamg_decide_peer_smoothers)
        // (at most 15 sweeps per launch: the row stamps of the kernel are bytes)
        for (int left = nSweeps; left > 0;)
        {
            const int kk = left > 15 ? 15 : left;
            const int rc = k_sweep_gs_wg_peer(a, kk, psi, source, m->d_diag, m->d_valA, m->d_bou, a->d_cycPair);
            if (rc > 0) { ldu_set_error("one-launch smoother: refused by an addressing that was declared eligible"); return -1; }
            if (rc < 0) return rc;
            a->ctx->nHaloExchanges += kk;
            a->ctx->nHaloOverlapped += kk;
            left -= kk;
        }
        return 0;
    }
    // bPrime differs from the source in the boundary rows only: one copy per call, then the boundary rows are
    // rewritten from the source before every sweep (one small kernel per sweep instead of copy + apply)
    if (a->nPatchFaces && !sym && nSweeps > 0 && k_ew(a->nCells, EW_COPY, bPrime, source, nullptr, s)) return -1;
    for (int sweep = 0; sweep < nSweeps; sweep++)
    {
        const double* rhs = source;
        if (a->nPatchFaces)
        {
            // bPrime = source; coupled boundaries Jacobi-style with negated coefficients
            // (GaussSeidelSmoother.C:98-145)
            if (dev_halo_start(m, psi)) return -1;
            if (sym)
            {
                // (the symmetric smoother's forward sweep overwrites bPrime: full copy every sweep)
                if (k_ew(a->nCells, EW_COPY, bPrime, source, nullptr, s)) return -1;
                if (k_apply_patches(a, bPrime, m->d_bou, -1.0, s)) return -1;
            }
            else if (k_apply_patches_from(a, bPrime, source, m->d_bou, -1.0, s)) return -1;
            rhs = bPrime;
            if (!sym && a->ctx->sweepP2P)
            {
                // small matrix with coupled patches: the frozen interface terms are in bPrime, so the single-wavefront
                // kernel can take this sweep like any other right-hand side (one sweep per launch: the neighbour
                // values change between sweeps, GaussSeidelSmoother.C:98-145)
                int rc = k_sweep_gs_wg(a, 1, psi, rhs, m->d_diag, m->d_valA);
                if (rc > 0) rc = k_sweep_gs_small(a, 1, psi, rhs, m->d_diag, m->d_valA);
                if (rc < 0) return rc;
                if (rc == 0) continue;
            }
        }
        SweepArgs g{};
        g.mode = SW_GS_FWD; g.w = psi; g.rhs = rhs; g.scale = m->d_diag; g.val = m->d_valA;
        g.aux = sym ? bPrime : nullptr;
        if (k_sweep(a, g)) return -1;
        if (sym)
        {
            SweepArgs b{};
            b.mode = SW_GS_BWD; b.w = psi; b.rhs = bPrime; b.scale = m->d_diag; b.val = m->d_valA;
            if (k_sweep(a, b)) return -1;
        }
    }
    return 0;
}

// DICSmoother.C:67-116, DILUSmoother.C:67-119, FDICSmoother.C:98-146
static int smooth_dic(ldu_matrix* m, int kind, double* psi, const double* source, int nSweeps)
{
    ldu_addr* a = m->a;
    hipStream_t s = a->ctx->stream;
    if (matrix_ensure_rD(m, kind)) return -1;
    double* rA = m->workVec(13);
    double* w = m->workVec(14);
    for (int sweep = 0; sweep < nSweeps; sweep++)
    {
        if (dev_residual(m, rA, psi, source)) return -1;
        SweepArgs f{};
        f.mode = SW_TRI_FWD; f.w = w; f.rhs = rA; f.scale = m->d_rD; f.val = m->d_valP;
        if (k_sweep(a, f)) return -1;
        SweepArgs b{};
        b.mode = SW_TRI_BWD; b.w = w; b.val = m->d_valP;
        if (k_sweep(a, b)) return -1;
        if (k_ew(a->nCells, EW_ADD_INPLACE, psi, w, nullptr, s)) return -1;
    }
    return 0;
}

// The engine-side copies of this matrix's coefficients that its GaussSeidel smoothing calls will ask for (block layouts,
// cluster layout), filled on stream s - the coefficient chain of a GAMG solve - instead of by the first smoothing call on the
// main stream.  The choice of engine is gs_sweeps' own (same predicates); engines whose plans do not exist yet are left to
// the smoothing call.
int dev_smooth_prefill(ldu_matrix* m, int smoother, hipStream_t s)
{
    ldu_addr* a = m->a;
    if (smoother != LDU_SM_GAUSSSEIDEL && smoother != LDU_SM_NONBLOCKINGGAUSSSEIDEL) return 0;
    if (!a->ctx->sweepP2P || !m->d_valA) return 0;
    if (!a->nPatchFaces)
    {
        if (a->ctx->clusterMulti && k_cluster_active(a)) return k_cluster_prefill(a, m->d_valA, s) < 0 ? -1 : 0;
        if (k_blocks_active(a)) return k_blocks_prefill(a, m->d_valA, nullptr, s) < 0 ? -1 : 0;
        return 0;
    }
    if (smoother == LDU_SM_GAUSSSEIDEL && k_blocks_active(a)) return k_blocks_prefill(a, m->d_valA, m->d_bou, s) < 0 ? -1 : 0;
    return 0;
}

int dev_smooth(ldu_matrix* m, int smoother, double* psi, const double* source, int nSweeps)
{
    switch (smoother)
    {
    case LDU_SM_GAUSSSEIDEL:    return smooth_gs(m, psi, source, nSweeps, false);
    case LDU_SM_NONBLOCKINGGAUSSSEIDEL:
        // nonBlockingGaussSeidelSmoother.C:46-240 sweeps the cells below blockStart_ before the halo
        // arrives and the rest after it.  Without coupled patches there is nothing to wait for and the
        // cell loop is the GaussSeidel loop, operation for operation.  With coupled patches the halo
        // term enters bPrime between the two blocks (a different rounding order): level kernels.
        if (m->a->nPatchFaces)
        {
            for (int sweep = 0; sweep < nSweeps; sweep++)
            {
                if (dev_halo_start(m, psi)) return -1;
                if (k_sweep_gs_nonblocking(m->a, psi, source, m->d_diag, m->d_valA, m->d_bou)) return -1;
            }
            return 0;
        }
        return smooth_gs(m, psi, source, nSweeps, false);
    case LDU_SM_SYMGAUSSSEIDEL: return smooth_gs(m, psi, source, nSweeps, true);
    case LDU_SM_DIC:
    case LDU_SM_FDIC:           return smooth_dic(m, LDU_PRE_DIC, psi, source, nSweeps);
    case LDU_SM_DILU:           return smooth_dic(m, LDU_PRE_DILU, psi, source, nSweeps);
    case LDU_SM_DICGAUSSSEIDEL:   // DICGaussSeidelSmoother.C:79-89
        if (smooth_dic(m, LDU_PRE_DIC, psi, source, nSweeps)) return -1;
        return smooth_gs(m, psi, source, nSweeps, false);
    case LDU_SM_DILUGAUSSSEIDEL:
        if (smooth_dic(m, LDU_PRE_DILU, psi, source, nSweeps)) return -1;
        return smooth_gs(m, psi, source, nSweeps, false);
    }
    ldu_set_error("unknown smoother");
    return -3;
}

// ---------------------------------------------------------------- convergence helpers

// SolverPerformance.C:59-91
static bool check_convergence(ldu_perf* p, double tol, double relTol)
{
    p->converged = (p->finalResidual < tol
                    || (relTol > kSmall && p->finalResidual < relTol * p->initialResidual)) ? 1 : 0;
    return p->converged != 0;
}

static void hist_push(ldu_perf* p, const ldu_controls* c, double* hist)
{
    if (hist && p->nHistory < c->historyCapacity) hist[p->nHistory] = p->finalResidual;
    p->nHistory++;
}

// normFactor (lduMatrixSolver.C:179-197) -> device scalar S_NORM; leaves sumA*avg nowhere
// (computed on the fly).  tmp receives sumA.
static int dev_normFactor(ldu_matrix* m, const double* psi, const double* source, const double* Apsi,
                          double* tmp)
{
    ldu_addr* a = m->a;
    ldu_ctx* ctx = a->ctx;
    hipStream_t s = ctx->stream;
    if (dev_sumA(m, tmp)) return -1;
    // gAverage(psi) = allreduce(sum)/allreduce(n) (FieldFunctions.C:514-533)
    if (k_reduce(ctx, a->nCells, RED_SUM, psi, nullptr, nullptr, nullptr, S_SUMPSI, s)) return -1;
    {
        const double cnt = (double)a->nCells;
        LDU_CHECK_HIP(hipMemcpyAsync(ctx->S() + S_COUNT, &cnt, sizeof(double), hipMemcpyHostToDevice, s));
        LDU_CHECK_HIP(hipStreamSynchronize(s));
    }
    if (comm_allreduce_scalars(ctx, S_SUMPSI, 1, s)) return -1;
    if (comm_allreduce_scalars(ctx, S_COUNT, 1, s)) return -1;
    if (k_reduce(ctx, a->nCells, RED_NORMFACTOR, Apsi, source, tmp, nullptr, S_NORM, s)) return -1;
    if (comm_allreduce_scalars(ctx, S_NORM, 1, s)) return -1;
    // + solverPerformance::small_
    double nf;
    if (dev_read_scalars(ctx, S_NORM, 1, &nf)) return -1;
    nf += kSmall;
    return dev_write_scalar(ctx, S_NORM, nf);
}

// ---------------------------------------------------------------- PCG / PBiCG

// An error return out of the Krylov loop may leave work forked onto the second stream (PBiCG's transposed system):
// join it before the caller sees the error, so that nothing of this solve is still running behind its back.
static int krylov_joined(ldu_matrix* m, int rc)
{
    if (rc < 0)
    {
        ldu_ctx* ctx = m->a->ctx;
        (void)hipStreamSynchronize(ctx->stream2);
        ctx->dualActive = 0;
    }
    return rc;
}

static int solve_krylov(ldu_matrix* m, const ldu_controls* c, double* psi, const double* source,
                        ldu_perf* perf, double* hist, bool bi)
{
    ldu_addr* a = m->a;
    ldu_ctx* ctx = a->ctx;
    hipStream_t s = ctx->stream;
    const int n = a->nCells;
    double* pA = m->workVec(0);
    double* wA = m->workVec(1);
    double* rA = m->workVec(2);
    double *pT = nullptr, *wT = nullptr, *rT = nullptr;
    if (bi) { pT = m->workVec(3); wT = m->workVec(4); rT = m->workVec(5); }

    // --- A.psi, initial residual, normalisation factor (PCG.C:93-108)
    if (dev_amul(m, wA, psi, false)) return -1;
    if (k_ew(n, EW_SUB, rA, source, wA, s)) return -1;
    if (bi)
    {
        if (dev_amul(m, wT, psi, true)) return -1;
        if (k_ew(n, EW_SUB, rT, source, wT, s)) return -1;
        if (k_ew(n, EW_ZERO, pT, nullptr, nullptr, s)) return -1;
    }
    if (dev_normFactor(m, psi, source, wA, pA)) return -1;
    if (k_reduce(ctx, n, RED_SUMMAG, rA, nullptr, nullptr, nullptr, S_RES, s)) return -1;
    if (comm_allreduce_scalars(ctx, S_RES, 1, s)) return -1;
    {
        double z = 0.0;
        LDU_CHECK_HIP(hipMemcpyAsync(ctx->S() + S_SINGULAR, &z, sizeof(double), hipMemcpyHostToDevice, s));
        LDU_CHECK_HIP(hipMemcpyAsync(ctx->S() + S_STOP, &z, sizeof(double), hipMemcpyHostToDevice, s));
        double g[2] = {kGreat, kGreat};   // wArA = great_ (PCG.C:89)
        LDU_CHECK_HIP(hipMemcpyAsync(ctx->S() + S_WARA0, g, 2 * sizeof(double), hipMemcpyHostToDevice, s));
        LDU_CHECK_HIP(hipStreamSynchronize(s));
    }
    double v[2];
    if (dev_read_scalars(ctx, S_RES, 2, v)) return -1;   // S_RES, S_NORM adjacent
    const double normFactor = v[1];
    perf->normFactor = normFactor;
    perf->initialResidual = v[0] / normFactor;
    perf->finalResidual = perf->initialResidual;
    hist_push(perf, c, hist);
    // The convergence read-back off the critical path: iteration k + 1 is queued BEFORE the residual of iteration k is read.
    // Everything an iteration writes before its last kernel is work space (wA, pA, the scalar slots); psi and rA change in
    // k_pcg_update_xr only, and that kernel does nothing once the device itself has found the loop condition of
    // PCG.C:174-181 false (k_krylov_decide: the host's own test on the same doubles).  Iterations, psi and the residual
    // history are those of the sequential loop; the price is one iteration of discarded work per solve.  Not with the GAMG
    // preconditioner (it reads scalars back itself).
    const bool speculate = ctx->krylovSpeculate && c->preconditioner != LDU_PRE_GAMG && c->maxIter > 0;
    if (speculate)
    {
        const double ir = perf->initialResidual;
        LDU_CHECK_HIP(hipMemcpyAsync(ctx->S() + S_INIT, &ir, sizeof(double), hipMemcpyHostToDevice, s));
        LDU_CHECK_HIP(hipStreamSynchronize(s));
    }

    if (!check_convergence(perf, c->tolerance, c->relTol))
    {
        const int pre = c->preconditioner;
        if (pre == LDU_PRE_GAMG)
        {
            if (gamg_precondition_setup(m, c)) return -1;
        }
        else if (pre != LDU_PRE_NONE)
        {
            if (matrix_ensure_rD(m, pre)) return -1;
        }
        int cur = S_WARA0, prev = S_WARA1;
        const bool dual = bi && !a->nPatchFaces && ctx->dualStream && pre != LDU_PRE_GAMG;
        // one iteration, queued: PCG.C:123-172 / PBiCG.C:130-179 up to the residual sum (S_RES)
        auto iteration = [&](int it) -> int {
            // wArAold = wArA (slot swap)
            { int t = cur; cur = prev; prev = t; }
            // --- precondition (PCG.C:129 / PBiCG.C:138-139)
            if (pre == LDU_PRE_GAMG)
            {
                if (gamg_precondition(m, c, wA, rA)) return -1;
            }
            else
            {
                // PBiCG: the transposed system is independent until the dot product -> second
                // stream, own point-to-point lane (both are latency-bound: they overlap fully)
                ctx->dualActive = dual ? 1 : 0;
                if (dual)
                {
                    LDU_CHECK_HIP(hipEventRecord(ctx->evFork, s));
                    LDU_CHECK_HIP(hipStreamWaitEvent(ctx->stream2, ctx->evFork, 0));
                    if (dev_precondition(m, pre, wT, rT, true, ctx->stream2)) { ctx->dualActive = 0; return -1; }
                    LDU_CHECK_HIP(hipEventRecord(ctx->evJoin, ctx->stream2));
                }
                const int rcA = dev_precondition(m, pre, wA, rA, false, s);
                ctx->dualActive = 0;
                if (rcA) return -1;
                if (bi && !dual && dev_precondition(m, pre, wT, rT, true, s)) return -1;
                if (dual) LDU_CHECK_HIP(hipStreamWaitEvent(s, ctx->evJoin, 0));
            }
            // --- wArA = gSumProd(wA, rA) / wArT = gSumProd(wA, rT)
            if (k_reduce(ctx, n, RED_DOT, wA, bi ? rT : rA, nullptr, nullptr, cur, s)) return -1;
            if (comm_allreduce_scalars(ctx, cur, 1, s)) return -1;
            // --- search directions
            const int first = it == 0;
            if (bi)
            {
                if (k_pbicg_update_p(n, pA, wA, pT, wT, ctx->S(), cur, prev, first, s)) return -1;
            }
            else if (k_pcg_update_p(n, pA, wA, ctx->S(), cur, prev, first, s)) return -1;
            // --- wA = A pA (wT = T pT)
            if (dual)
            {
                LDU_CHECK_HIP(hipEventRecord(ctx->evFork, s));
                LDU_CHECK_HIP(hipStreamWaitEvent(ctx->stream2, ctx->evFork, 0));
                if (dev_amul(m, wT, pT, true, ctx->stream2)) return -1;
                LDU_CHECK_HIP(hipEventRecord(ctx->evJoin, ctx->stream2));
            }
            if (dev_amul(m, wA, pA, false)) return -1;
            if (bi && !dual && dev_amul(m, wT, pT, true)) return -1;
            if (dual) LDU_CHECK_HIP(hipStreamWaitEvent(s, ctx->evJoin, 0));
            // --- wApA = gSumProd(wA, pA) / wApT = gSumProd(wA, pT)
            if (k_reduce(ctx, n, RED_DOT, wA, bi ? pT : pA, nullptr, nullptr, S_WAPA, s)) return -1;
            if (comm_allreduce_scalars(ctx, S_WAPA, 1, s)) return -1;
            // --- singularity test + psi/rA update + |rA| partial sums, one pass
            if (k_pcg_update_xr(ctx, n, psi, rA, pA, wA, rT, wT, cur, s)) return -1;
            if (comm_allreduce_scalars(ctx, S_RES, 1, s)) return -1;
            return 0;
        };
        if (!speculate)
        {
            do
            {
                if (iteration(perf->nIterations)) return -1;
                double rs[3];
                if (dev_read_scalars(ctx, S_RES, 3, rs)) return -1;   // S_RES, S_NORM, S_SINGULAR
                if (rs[2] != 0.0) { perf->singular = 1; break; }
                perf->finalResidual = rs[0] / normFactor;
                hist_push(perf, c, hist);
            } while (perf->nIterations++ < c->maxIter && !check_convergence(perf, c->tolerance, c->relTol));
        }
        else
        {
            // queue(it): iteration it, the device-side loop condition, the scalars on their way to the host
            auto queue = [&](int it) -> int {
                if (iteration(it)) return -1;
                if (k_krylov_decide(ctx, c->tolerance, c->relTol, it, c->maxIter, s)) return -1;
                if (comm_allreduce_abort(ctx, s)) return -1;
                ctx->nScalarReadbacks++;
                LDU_CHECK_HIP(hipMemcpyAsync(ctx->h_ring[it & 1], ctx->d_scalars, sizeof(double) * (S_NSLOTS + 2),
                                             hipMemcpyDeviceToHost, s));
                LDU_CHECK_HIP(hipEventRecord(ctx->evRing[it & 1], s));
                return 0;
            };
            int it = 0;
            bool ahead = false;      // iteration it + 1 is in the queue
            if (queue(0)) return -1;
            for (;;)
            {
                // iteration it + 1 goes into the queue while iteration it runs (the reference's loop condition allows it
                // for it < maxIter; whether it takes effect is the device's decision)
                const bool more = it < c->maxIter;
                if (more && queue(it + 1)) return -1;
                ahead = more;
                LDU_CHECK_HIP(hipEventSynchronize(ctx->evRing[it & 1]));
                const double* H = ctx->h_ring[it & 1];
                {
                    const int rcA = abort_words_to_error(ctx, (const int*)(H + S_NSLOTS), s);
                    if (rcA) return rcA;
                }
                const double* rs = H + ctx->sb + S_RES;              // S_RES, S_NORM, S_SINGULAR
                if (rs[2] != 0.0) { perf->singular = 1; break; }
                perf->finalResidual = rs[0] / normFactor;
                hist_push(perf, c, hist);
                if (!(perf->nIterations++ < c->maxIter && !check_convergence(perf, c->tolerance, c->relTol))) break;
                it++;
            }
            if (ahead)
            {
                // The iteration queued behind the last one is discarded work (its update kernel does nothing), but it must
                // not outlive this call, and what it may have left in the flag words must not surface in the NEXT operation:
                // a peer time-out is an error of this solve (the ranks are out of step); a sweep of discarded work that gave
                // up waiting is dropped - words cleared, a new tag generation for the granules it left half written.
                LDU_CHECK_HIP(hipEventSynchronize(ctx->evRing[(it + 1) & 1]));
                const int* w = (const int*)(ctx->h_ring[(it + 1) & 1] + S_NSLOTS);
                if (w[2]) return abort_words_to_error(ctx, w, s);
                if (w[0] || w[1])
                {
                    LDU_CHECK_HIP(hipMemsetAsync(ctx->d_abort, 0, 2 * sizeof(int), s));
                    ctx->p2pGen++;
                    ctx->nDiscardedAborts++;
                }
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------- smoothSolver

static int solve_smooth(ldu_matrix* m, const ldu_controls* c, double* psi, const double* source,
                        ldu_perf* perf, double* hist)
{
    ldu_addr* a = m->a;
    ldu_ctx* ctx = a->ctx;
    hipStream_t s = ctx->stream;
    const int n = a->nCells;
    if (c->nSweeps < 0)
    {
        if (dev_smooth(m, c->smoother, psi, source, -c->nSweeps)) return -1;
        perf->nIterations -= c->nSweeps;
        return 0;
    }
    double* Apsi = m->workVec(0);
    double* temp = m->workVec(1);
    if (dev_amul(m, Apsi, psi, false)) return -1;
    if (dev_normFactor(m, psi, source, Apsi, temp)) return -1;
    if (k_ew(n, EW_SUB, temp, source, Apsi, s)) return -1;
    if (k_reduce(ctx, n, RED_SUMMAG, temp, nullptr, nullptr, nullptr, S_RES, s)) return -1;
    if (comm_allreduce_scalars(ctx, S_RES, 1, s)) return -1;
    double v[2];
    if (dev_read_scalars(ctx, S_RES, 2, v)) return -1;
    const double normFactor = v[1];
    perf->normFactor = normFactor;
    perf->initialResidual = v[0] / normFactor;
    perf->finalResidual = perf->initialResidual;
    hist_push(perf, c, hist);
    if (!check_convergence(perf, c->tolerance, c->relTol))
    {
        do
        {
            if (dev_smooth(m, c->smoother, psi, source, c->nSweeps)) return -1;
            if (dev_residual(m, temp, psi, source)) return -1;
            if (k_reduce(ctx, n, RED_SUMMAG, temp, nullptr, nullptr, nullptr, S_RES, s)) return -1;
            if (comm_allreduce_scalars(ctx, S_RES, 1, s)) return -1;
            double r;
            if (dev_read_scalars(ctx, S_RES, 1, &r)) return -1;
            perf->finalResidual = r / normFactor;
            hist_push(perf, c, hist);
        } while ((perf->nIterations += c->nSweeps) < c->maxIter
                 && !check_convergence(perf, c->tolerance, c->relTol));
    }
    return 0;
}

int dev_solve(ldu_matrix* m, const ldu_controls* c, double* psi, const double* source, ldu_perf* perf,
              double* hist)
{
    ldu_addr* a = m->a;
    switch (c->solver)
    {
    case LDU_SOLVER_PCG:
        if (!m->sym)
        {
            // PCG is registered in the symmetric table only (PCG.C:34-35): on an asymmetric matrix the reference
            // aborts in lduMatrix::solver::New (lduMatrixSolver.C:96-110)
            ldu_set_error("Unknown asymmetric matrix solver PCG (lduMatrixSolver.C:96-110)");
            return -16;
        }
        return krylov_joined(m, solve_krylov(m, c, psi, source, perf, hist, false));
    case LDU_SOLVER_PBICG:
        return krylov_joined(m, solve_krylov(m, c, psi, source, perf, hist, true));
    case LDU_SOLVER_SMOOTH:
        return solve_smooth(m, c, psi, source, perf, hist);
    case LDU_SOLVER_GAMG:
        return gamg_solve(m, c, psi, source, perf, hist);
    case LDU_SOLVER_DIAGONAL:   // diagonalSolver.C:62-81
        if (k_ew(a->nCells, EW_DIV, psi, source, m->d_diag, a->ctx->stream)) return -1;
        perf->converged = 1;
        return 0;
    }
    ldu_set_error("unknown solver");
    return -3;
}
