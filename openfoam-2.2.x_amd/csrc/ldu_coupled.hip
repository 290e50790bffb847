// Coupled solvers: the templated LduMatrix<Type, scalar, scalar> family of the reference
// (src/OpenFOAM/matrices/LduMatrix: PCICG, PBiCCCG, PBiCICG, SmoothSolver, DiagonalSolver, TDILU, diagonal and
// no preconditioner, TGaussSeidel smoother) on the coefficients of an ldu_matrix.
//
// Device layout: a Field<Type> is held component-major - nCmpt planes of `stride` doubles in the level-ordered
// numbering of the plan - so every plane is an ordinary scalar vector for the sweep engines, and the row
// kernels (Amul / Tmul / residual) read each coefficient and column index once for all components.
// The family associates products differently from the lduMatrix solvers (rD*(coeff*x) instead of
// (rD*coeff)*x, a reciprocal multiply instead of a division): sweep modes SW_*_T of the engines.
// Scalars (wArA, alpha, ...) are per component; they are read back once per reduction - the residual has to
// reach the host for checkConvergence anyway.
#include <chrono>
#include <cmath>
#include <cstring>

#include "ldu_internal.hpp"

#define BLK 256
#define WPB (BLK / LDU_WAVE)
#define CR_MAXG 512          // partial sums per component
#define CSLOT 96             // ctx scalar slots [CSLOT, CSLOT + 9): outside every solver bank

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int ewGrid(long n) { int g = cdiv(n, BLK); return g < 1 ? 1 : (g > 2048 ? 2048 : g); }

static const double kGreat = 1e20;    // SolverPerformance.H
static const double kSmall = 1e-20;
static const double kVSmall = 1e-300;

struct CoupledWork {
    int nc = 0;
    size_t stride = 0;
    std::vector<double*> fields;
    double* d_rDT = nullptr;       // TDILU reciprocal preconditioned diagonal
    uint64_t rDTEpoch = ~0ull;
    double* d_partials = nullptr;  // [9][CR_MAXG]
    double* d_stage[3] = {nullptr, nullptr, nullptr};   // interleaved images of host buffers
    size_t stageDoubles = 0;
};

void coupled_free(ldu_matrix* m)
{
    CoupledWork* W = m->coupled;
    if (!W) return;
    for (double* p : W->fields) if (p) (void)hipFree(p);
    if (W->d_rDT) (void)hipFree(W->d_rDT);
    if (W->d_partials) (void)hipFree(W->d_partials);
    for (double* p : W->d_stage) if (p) (void)hipFree(p);
    delete W;
    m->coupled = nullptr;
}

void coupled_invalidate(ldu_matrix* m)
{
    if (m->coupled) m->coupled->rDTEpoch = ~0ull;
}

static CoupledWork* work_of(ldu_matrix* m, int nc)
{
    if (!m->coupled) m->coupled = new CoupledWork;
    CoupledWork* W = m->coupled;
    const size_t stride = ((size_t)m->a->nCells + 127) / 64 * 64;
    if (W->nc != nc || W->stride != stride)
    {
        for (double*& p : W->fields) { if (p) (void)hipFree(p); p = nullptr; }
        W->nc = nc;
        W->stride = stride;
    }
    if (!W->d_partials && hipMalloc((void**)&W->d_partials, sizeof(double) * LDU_MAX_CMPT * CR_MAXG) != hipSuccess)
        return nullptr;
    return W;
}

static double* field(CoupledWork* W, int i)
{
    while ((int)W->fields.size() <= i) W->fields.push_back(nullptr);
    if (!W->fields[i])
        if (hipMalloc((void**)&W->fields[i], sizeof(double) * W->stride * (size_t)W->nc) != hipSuccess) return nullptr;
    return W->fields[i];
}

static bool on_device(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// ---------------------------------------------------------------- layout
// Field<Type> (cell-major, original order) <-> component planes in the level-ordered numbering
__global__ void cmpt_gather_kernel(int n, int nc, size_t stride, const int* __restrict__ perm,
                                   const double* __restrict__ src, double* __restrict__ dst)
{
    const int c = blockIdx.y;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
        dst[(size_t)c * stride + i] = src[(size_t)perm[i] * nc + c];
}
__global__ void cmpt_scatter_kernel(int n, int nc, size_t stride, const int* __restrict__ perm,
                                    const double* __restrict__ src, double* __restrict__ dst)
{
    const int c = blockIdx.y;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
        dst[(size_t)perm[i] * nc + c] = src[(size_t)c * stride + i];
}

static int stage_grow(CoupledWork* W, size_t doubles)
{
    if (W->stageDoubles >= doubles) return 0;
    for (double*& p : W->d_stage) { if (p) (void)hipFree(p); p = nullptr; }
    for (double*& p : W->d_stage) LDU_CHECK_HIP(hipMalloc((void**)&p, sizeof(double) * doubles));
    W->stageDoubles = doubles;
    return 0;
}

// user Field<Type> -> planes (work field `fi`)
static double* field_in(ldu_matrix* m, CoupledWork* W, int fi, int stageI, const double* user)
{
    ldu_addr* a = m->a;
    hipStream_t s = a->ctx->stream;
    double* f = field(W, fi);
    if (!f) return nullptr;
    if (!a->nCells) return f;
    const double* dev = user;
    if (!on_device(user))
    {
        if (stage_grow(W, (size_t)a->nCells * W->nc + 64)) return nullptr;
        if (hipMemcpyAsync(W->d_stage[stageI], user, sizeof(double) * (size_t)a->nCells * W->nc, hipMemcpyHostToDevice,
                           s) != hipSuccess) return nullptr;
        dev = W->d_stage[stageI];
    }
    cmpt_gather_kernel<<<dim3(ewGrid(a->nCells), W->nc), BLK, 0, s>>>(a->nCells, W->nc, W->stride, a->d_perm, dev, f);
    if (hipGetLastError() != hipSuccess) return nullptr;
    return f;
}

static int field_out(ldu_matrix* m, CoupledWork* W, double* user, const double* f)
{
    ldu_addr* a = m->a;
    hipStream_t s = a->ctx->stream;
    if (!a->nCells) return 0;
    double* dev = user;
    const bool host = !on_device(user);
    if (host)
    {
        if (stage_grow(W, (size_t)a->nCells * W->nc + 64)) return -1;
        dev = W->d_stage[2];
    }
    cmpt_scatter_kernel<<<dim3(ewGrid(a->nCells), W->nc), BLK, 0, s>>>(a->nCells, W->nc, W->stride, a->d_perm, f, dev);
    LDU_CHECK_HIP(hipGetLastError());
    if (host)
        LDU_CHECK_HIP(hipMemcpyAsync(user, dev, sizeof(double) * (size_t)a->nCells * W->nc, hipMemcpyDeviceToHost, s));
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

// ---------------------------------------------------------------- row kernels, up to three planes per pass
// MODE 0: y = diag*x + sum val*x[col]   (LduMatrixATmul.C:88-103 / :139-154)
// MODE 1: y = b - diag*x - sum val*x[col]   (:250-265)
// Per plane the operations and their order are those of the scalar row kernel.
template <int MODE>
__global__ void __launch_bounds__(BLK)
crow_kernel(int nSlices, const int* __restrict__ sliceRow, const int* __restrict__ sliceCnt,
            const int* __restrict__ sliceEnt, const int* __restrict__ sliceW,
            const unsigned char* __restrict__ nL, const unsigned char* __restrict__ nU,
            const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ diag,
            const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ y, size_t stride, int nc)
{
    const int s = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (s >= nSlices) return;
    const int lane = threadIdx.x & 63;
    if (lane >= sliceCnt[s]) return;
    const int r = sliceRow[s] + lane;
    const int n = (int)nL[r] + (int)nU[r];
    const long ent = (long)sliceEnt[s] + lane;
    const int c0 = blockIdx.y * 3;
    const int cn = nc - c0 < 3 ? nc - c0 : 3;
    const double d = diag[r];
    double acc[3];
#pragma unroll
    for (int j = 0; j < 3; j++)
    {
        acc[j] = 0.0;
        if (j < cn)
        {
            const size_t o = (size_t)(c0 + j) * stride + r;
            acc[j] = MODE == 0 ? d * x[o] : b[o] - d * x[o];
        }
    }
    const int W = sliceW[s];
    if (W <= 8)
    {
        int c[8];
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < W)
            {
                const long e = ent + (long)k * LDU_WAVE;
                c[k] = col[e];
                v[k] = val[e];
            }
#pragma unroll
        for (int j = 0; j < 3; j++)
        {
            if (j >= cn) continue;
            const double* xp = x + (size_t)(c0 + j) * stride;
            double xv[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < W) xv[k] = xp[c[k]];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < W && k < n)
                {
                    if (MODE == 0) acc[j] += v[k] * xv[k];
                    else acc[j] -= v[k] * xv[k];
                }
        }
    }
    else
    {
        for (int k = 0; k < n; k++)
        {
            const long e = ent + (long)k * LDU_WAVE;
            const double v = val[e];
            const int cc = col[e];
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (j < cn)
                {
                    const double xv = x[(size_t)(c0 + j) * stride + cc];
                    if (MODE == 0) acc[j] += v * xv;
                    else acc[j] -= v * xv;
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (j < cn) y[(size_t)(c0 + j) * stride + r] = acc[j];
}

template <int MODE>
static int launch_crow(ldu_matrix* m, CoupledWork* W, double* y, const double* x, const double* b, const double* val)
{
    ldu_addr* a = m->a;
    if (a->nSlices == 0) return 0;
    crow_kernel<MODE><<<dim3(cdiv(a->nSlices, WPB), (W->nc + 2) / 3), BLK, 0, a->ctx->stream>>>(a->nSlices,
        a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_sliceW, a->d_nL, a->d_nU, a->d_col, val, m->d_diag, x, b, y,
        W->stride, W->nc);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// Amul / Tmul with the interfaces (LduMatrixATmul.C:66-165): rows first, then every plane's coupled faces
static int c_amul(ldu_matrix* m, CoupledWork* W, double* y, const double* x, bool transpose)
{
    ldu_addr* a = m->a;
    if (launch_crow<0>(m, W, y, x, nullptr, transpose ? m->d_valT : m->d_valA)) return -1;
    if (a->nPatchFaces)
        for (int c = 0; c < W->nc; c++)
        {
            if (dev_halo_start(m, x + c * W->stride)) return -1;
            if (k_apply_patches(a, y + c * W->stride, transpose ? m->d_int : m->d_bou, 1.0, a->ctx->stream)) return -1;
        }
    return 0;
}

// residual (LduMatrixATmul.C:218-276): negated interface coefficients
static int c_residual(ldu_matrix* m, CoupledWork* W, double* r, const double* x, const double* b)
{
    ldu_addr* a = m->a;
    if (launch_crow<1>(m, W, r, x, b, m->d_valA)) return -1;
    if (a->nPatchFaces)
        for (int c = 0; c < W->nc; c++)
        {
            if (dev_halo_start(m, x + c * W->stride)) return -1;
            if (k_apply_patches(a, r + c * W->stride, m->d_bou, -1.0, a->ctx->stream)) return -1;
        }
    return 0;
}

// ---------------------------------------------------------------- elementwise, one plane per blockIdx.y
struct Cmpts { double v[LDU_MAX_CMPT]; };
enum { CEW_COPY = 0,        // y = a
       CEW_SUB = 1,         // y = a - b
       CEW_MUL_S = 2,       // y = s*a            (s: one scalar plane, e.g. rD)
       CEW_DIV_S = 3,       // y = a/s
       CEW_P_UPDATE = 4,    // y = a + k_c*y      (pA = wA + beta*pA)
       CEW_ADD_K = 5,       // y += k_c*a
       CEW_SUB_K = 6,       // y -= k_c*a
       CEW_ZERO = 7,
       CEW_ADD = 8 };       // y += a
template <int OP>
__global__ void __launch_bounds__(BLK)
cew_kernel(int n, size_t stride, double* __restrict__ y, const double* __restrict__ a, const double* __restrict__ b,
           Cmpts k)
{
    const int c = blockIdx.y;
    const size_t o = (size_t)c * stride;
    const double kc = k.v[c];
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
    {
        if (OP == CEW_COPY) y[o + i] = a[o + i];
        else if (OP == CEW_SUB) y[o + i] = a[o + i] - b[o + i];
        else if (OP == CEW_MUL_S) y[o + i] = b[i] * a[o + i];
        else if (OP == CEW_DIV_S) y[o + i] = a[o + i] / b[i];
        else if (OP == CEW_P_UPDATE) y[o + i] = a[o + i] + kc * y[o + i];
        else if (OP == CEW_ADD_K) y[o + i] += kc * a[o + i];
        else if (OP == CEW_SUB_K) y[o + i] -= kc * a[o + i];
        else if (OP == CEW_ZERO) y[o + i] = 0.0;
        else y[o + i] += a[o + i];
    }
}
template <int OP>
static int cew(ldu_matrix* m, CoupledWork* W, double* y, const double* a, const double* b, const Cmpts* k = nullptr)
{
    const int n = m->a->nCells;
    if (!n) return 0;
    Cmpts kk{};
    if (k) kk = *k;
    cew_kernel<OP><<<dim3(ewGrid(n), W->nc), BLK, 0, m->a->ctx->stream>>>(n, W->stride, y, a, b, kk);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- reductions, deterministic two-stage tree
enum { CR_DOT = 0,       // sum a_c*b_c                       (gSumCmptProd)
       CR_SUMMAG = 1,    // sum |a_c|                         (gSumCmptMag)
       CR_SUM = 2,       // sum a_c                           (gSum, for gAverage)
       CR_NORM = 3,      // sum |a_c - s*k_c| + |b_c - s*k_c| (normFactor, LduMatrixSolver.C:167-186)
       CR_DOTALL = 4 };  // sum_cells (a & b)                 (gSumProd of Field<Type>: one scalar)
template <int OP>
__global__ void __launch_bounds__(BLK)
cred_partial_kernel(int n, int nc, size_t stride, const double* __restrict__ a, const double* __restrict__ b,
                    const double* __restrict__ sPlane, Cmpts k, double* __restrict__ partials)
{
    __shared__ double sh[WPB];
    const int c = blockIdx.y;
    const size_t o = (size_t)c * stride;
    double acc = 0.0;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
    {
        if (OP == CR_DOT) acc += a[o + i] * b[o + i];
        else if (OP == CR_SUMMAG) acc += fabs(a[o + i]);
        else if (OP == CR_SUM) acc += a[o + i];
        else if (OP == CR_NORM)
        {
            const double t = sPlane[i] * k.v[c];
            acc += fabs(a[o + i] - t) + fabs(b[o + i] - t);
        }
        else
        {
            // Type && Type: weight*a*b per component, summed in component order (k.v = the weights)
            double d = (k.v[0] * a[i]) * b[i];
            for (int j = 1; j < nc; j++) d += (k.v[j] * a[(size_t)j * stride + i]) * b[(size_t)j * stride + i];
            acc += d;
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t = 0.0;
        for (int w = 0; w < WPB; w++) t += sh[w];
        partials[(size_t)c * CR_MAXG + blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(BLK)
cred_final_kernel(int g, const double* __restrict__ partials, double* __restrict__ out)
{
    __shared__ double sh[WPB];
    const int c = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < g; i += BLK) acc += partials[(size_t)c * CR_MAXG + i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t = 0.0;
        for (int w = 0; w < WPB; w++) t += sh[w];
        out[c] = t;
    }
}

// result -> host (all ranks' sum); also surfaces an aborted sweep
template <int OP>
static int cred(ldu_matrix* m, CoupledWork* W, const double* a, const double* b, const double* sPlane,
                const Cmpts* k, double* out)
{
    ldu_ctx* ctx = m->a->ctx;
    hipStream_t s = ctx->stream;
    const int n = m->a->nCells;
    const int planes = OP == CR_DOTALL ? 1 : W->nc;
    int g = cdiv(n, BLK * 8);
    if (g < 1) g = 1;
    if (g > CR_MAXG) g = CR_MAXG;
    Cmpts kk{};
    if (k) kk = *k;
    cred_partial_kernel<OP><<<dim3(g, planes), BLK, 0, s>>>(n, W->nc, W->stride, a, b, sPlane, kk, W->d_partials);
    cred_final_kernel<<<planes, BLK, 0, s>>>(g, W->d_partials, ctx->d_scalars + CSLOT);
    LDU_CHECK_HIP(hipGetLastError());
    if (comm_allreduce_scalars(ctx, CSLOT - ctx->sb, planes, s)) return -1;
    return dev_read_scalars(ctx, CSLOT - ctx->sb, planes, out);
}

// ---------------------------------------------------------------- preconditioners / smoother
// TDILUPreconditioner.C:46-79: rD[u] -= (upper*lower)*inv(rD[l]) in face order, then rD = inv(rD)
static int ensure_rDT(ldu_matrix* m, CoupledWork* W)
{
    ldu_addr* a = m->a;
    if (W->d_rDT && W->rDTEpoch == m->coeffEpoch) return 0;
    if (!W->d_rDT) LDU_CHECK_HIP(hipMalloc((void**)&W->d_rDT, sizeof(double) * (W->stride + 64)));
    SweepArgs g{};
    g.mode = SW_RD_T;
    g.w = W->d_rDT;
    g.scale = m->d_diag;
    g.val = m->d_valA;
    g.val2 = m->d_valT;
    if (k_sweep(a, g)) return -1;
    if (k_reciprocal(a->nCells, W->d_rDT, a->ctx->stream)) return -1;
    W->rDTEpoch = m->coeffEpoch;
    return 0;
}

static int c_precondition(ldu_matrix* m, CoupledWork* W, int pre, double* w, const double* r, bool transpose,
                          hipStream_t st = nullptr)
{
    ldu_addr* a = m->a;
    if (!st) st = a->ctx->stream;
    const int lane = st == a->ctx->stream2 ? 1 : 0;   // second concurrent sweep: own hand-off state
    switch (pre)
    {
    case LDU_CPRE_NONE:       // NoPreconditioner.C:49-56
        return cew<CEW_COPY>(m, W, w, r, nullptr);
    case LDU_CPRE_DIAGONAL:   // DiagonalPreconditioner.C:40-80: rD = inv(diag), wA = rD*rA
        if (matrix_ensure_rD(m, LDU_PRE_DIAGONAL)) return -1;
        return cew<CEW_MUL_S>(m, W, w, r, m->d_rDiag);
    case LDU_CPRE_DILU:
    {
        if (ensure_rDT(m, W)) return -1;
        // precondition: lower[] forward, upper[] backward (TDILUPreconditioner.C:108-124);
        // preconditionT: upper[] forward, lower[] backward (:154-175) = the transposed value array
        const double* val = (transpose && !m->sym) ? m->d_valT : m->d_valA;
        int c0 = 0;
        // three planes per cluster sweep: indices, coefficients and the waits are shared (ldu_cluster.hip)
        for (; c0 + 3 <= W->nc; c0 += 3)
        {
            double* wp = w + c0 * W->stride;
            int rc = k_sweep_cluster_vec3(a, SW_TRI_FWD_T, wp, r + c0 * W->stride, W->stride, W->d_rDT, val, lane, st);
            if (rc < 0) return -1;
            if (rc > 0) break;
            rc = k_sweep_cluster_vec3(a, SW_TRI_BWD_T, wp, nullptr, W->stride, W->d_rDT, val, lane, st);
            if (rc) return -1;
        }
        for (int c = c0; c < W->nc; c++)
        {
            SweepArgs f{};
            f.mode = SW_TRI_FWD_T; f.w = w + c * W->stride; f.rhs = r + c * W->stride; f.scale = W->d_rDT; f.val = val;
            f.lane = lane; f.stream = st;
            if (k_sweep(a, f)) return -1;
            SweepArgs b{};
            b.mode = SW_TRI_BWD_T; b.w = w + c * W->stride; b.scale = W->d_rDT; b.val = val;
            b.lane = lane; b.stream = st;
            if (k_sweep(a, b)) return -1;
        }
        return 0;
    }
    }
    ldu_set_error("unknown coupled preconditioner");
    return -3;
}

// TGaussSeidelSmoother.C:63-153: rD = inv(diag); per sweep bPrime = source (+ coupled faces, negated
// coefficients), then the row loop, each row finished with rD*curPsi
static int c_smooth(ldu_matrix* m, CoupledWork* W, double* psi, const double* source, int nSweeps)
{
    ldu_addr* a = m->a;
    hipStream_t s = a->ctx->stream;
    if (matrix_ensure_rD(m, LDU_PRE_DIAGONAL)) return -1;
    double* bPrime = a->nPatchFaces ? field(W, 10) : nullptr;
    if (a->nPatchFaces && !bPrime) return -1;
    for (int sweep = 0; sweep < nSweeps; sweep++)
    {
        int c0 = 0;
        if (!a->nPatchFaces)
            for (; c0 + 3 <= W->nc; c0 += 3)
            {
                const int rc = k_sweep_cluster_vec3(a, SW_GS_FWD_T, psi + c0 * W->stride, source + c0 * W->stride,
                                                    W->stride, m->d_rDiag, m->d_valA, 0, nullptr);
                if (rc < 0) return -1;
                if (rc > 0) break;
            }
        for (int c = c0; c < W->nc; c++)
        {
            const double* rhs = source + c * W->stride;
            if (a->nPatchFaces)
            {
                double* bp = bPrime + c * W->stride;
                if (dev_halo_start(m, psi + c * W->stride)) return -1;
                if (k_ew(a->nCells, EW_COPY, bp, rhs, nullptr, s)) return -1;
                if (k_apply_patches(a, bp, m->d_bou, -1.0, s)) return -1;
                rhs = bp;
            }
            SweepArgs g{};
            g.mode = SW_GS_FWD_T; g.w = psi + c * W->stride; g.rhs = rhs; g.scale = m->d_rDiag; g.val = m->d_valA;
            if (k_sweep(a, g)) return -1;
        }
    }
    return 0;
}

// ---------------------------------------------------------------- solver loops
static inline double stabilise(double x, double y) { return x < 0 ? x - y : x + y; }   // doubleScalar.H

// SolverPerformance.C:60-90 with the VectorSpace comparisons (VectorSpaceI.H:661-689: true iff every component)
static bool check_convergence(ldu_coupled_perf* p, const ldu_coupled_controls* c)
{
    const int nc = c->nCmpt;
    bool absOk = true, relOn = true, relOk = true;
    for (int i = 0; i < nc; i++)
    {
        absOk = absOk && p->finalResidual[i] < c->tolerance[i];
        relOn = relOn && c->relTol[i] > kSmall;
        relOk = relOk && p->finalResidual[i] < c->relTol[i] * p->initialResidual[i];
    }
    p->converged = (absOk || (relOn && relOk)) ? 1 : 0;
    return p->converged != 0;
}

// SolverPerformance.C:32-55
static bool check_singularity(ldu_coupled_perf* p, int nc, const double* wApA)
{
    bool all = true;
    for (int i = 0; i < nc; i++)
    {
        p->singular[i] = wApA[i] < kVSmall ? 1 : 0;
        all = all && p->singular[i];
    }
    return all;
}

// LduMatrixSolver.C:167-186; tmp receives nothing persistent
static int norm_factor(ldu_matrix* m, CoupledWork* W, const double* psi, const double* source, const double* Apsi,
                       double* nf)
{
    double* sumA = m->workVec(16);
    if (!sumA) return -1;
    if (dev_sumA(m, sumA)) return -1;
    // gAverage(psi): global sum over global count (FieldFunctions.C:514-533)
    double sum[LDU_MAX_CMPT];
    if (cred<CR_SUM>(m, W, psi, nullptr, nullptr, nullptr, sum)) return -1;
    ldu_ctx* ctx = m->a->ctx;
    double cnt = (double)m->a->nCells;
    {
        LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        LDU_CHECK_HIP(hipMemcpy(ctx->d_scalars + CSLOT + 10, &cnt, sizeof(double), hipMemcpyHostToDevice));
        if (comm_allreduce_scalars(ctx, CSLOT + 10 - ctx->sb, 1, ctx->stream)) return -1;
        if (dev_read_scalars(ctx, CSLOT + 10 - ctx->sb, 1, &cnt)) return -1;
    }
    Cmpts avg{};
    for (int c = 0; c < W->nc; c++) avg.v[c] = sum[c] / cnt;
    if (cred<CR_NORM>(m, W, Apsi, source, sumA, &avg, nf)) return -1;
    for (int c = 0; c < W->nc; c++) nf[c] = stabilise(nf[c], kSmall);
    return 0;
}

// PCICG.C:50-184 (bi = 0), PBiCICG.C:50-197 (bi = 1), PBiCCCG.C:50-192 (bi = 2: one scalar alpha / beta)
static int solve_krylov(ldu_matrix* m, CoupledWork* W, const ldu_coupled_controls* c, double* psi,
                        const double* source, ldu_coupled_perf* perf, int bi)
{
    const int nc = W->nc;
    double* pA = field(W, 2);
    double* wA = field(W, 3);
    double* rA = field(W, 4);
    double *pT = nullptr, *wT = nullptr, *rT = nullptr;
    if (bi) { pT = field(W, 5); wT = field(W, 6); rT = field(W, 7); }
    if (!pA || !wA || !rA || (bi && (!pT || !wT || !rT))) return -1;

    double wArA[LDU_MAX_CMPT], wArAold[LDU_MAX_CMPT], wApA[LDU_MAX_CMPT], res[LDU_MAX_CMPT];
    for (int i = 0; i < nc; i++) wArA[i] = bi == 2 ? 1e15 : kGreat;   // PBiCCCG.C:81
    Cmpts ipw{};
    for (int i = 0; i < nc; i++) ipw.v[i] = c->innerProductWeights[i];

    if (c_amul(m, W, wA, psi, false)) return -1;
    if (cew<CEW_SUB>(m, W, rA, source, wA)) return -1;
    if (bi)
    {
        if (cew<CEW_ZERO>(m, W, pT, nullptr, nullptr)) return -1;
        if (c_amul(m, W, wT, psi, true)) return -1;
        if (cew<CEW_SUB>(m, W, rT, source, wT)) return -1;
    }
    double* nf = perf->normFactor;
    if (norm_factor(m, W, psi, source, wA, nf)) return -1;
    if (cred<CR_SUMMAG>(m, W, rA, nullptr, nullptr, nullptr, res)) return -1;
    for (int i = 0; i < nc; i++) perf->initialResidual[i] = perf->finalResidual[i] = res[i] / nf[i];

    if (check_convergence(perf, c)) return 0;
    for (;;)
    {
        for (int i = 0; i < nc; i++) wArAold[i] = wArA[i];
        // the transposed system is independent until the dot product: second stream, own hand-off lane
        const bool dual = bi && c->preconditioner == LDU_CPRE_DILU && !m->a->nPatchFaces && m->a->ctx->dualStream;
        if (dual)
        {
            ldu_ctx* ctx = m->a->ctx;
            if (ensure_rDT(m, W)) return -1;
            LDU_CHECK_HIP(hipEventRecord(ctx->evFork, ctx->stream));
            LDU_CHECK_HIP(hipStreamWaitEvent(ctx->stream2, ctx->evFork, 0));
            ctx->dualActive = 1;
            if (c_precondition(m, W, c->preconditioner, wT, rT, true, ctx->stream2)) { ctx->dualActive = 0; return -1; }
            LDU_CHECK_HIP(hipEventRecord(ctx->evJoin, ctx->stream2));
            const int rcA = c_precondition(m, W, c->preconditioner, wA, rA, false);
            ctx->dualActive = 0;
            if (rcA) return -1;
            LDU_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->evJoin, 0));
        }
        else
        {
            if (c_precondition(m, W, c->preconditioner, wA, rA, false)) return -1;
            if (bi && c_precondition(m, W, c->preconditioner, wT, rT, true)) return -1;
        }
        if (bi == 2)
        {
            if (cred<CR_DOTALL>(m, W, wA, rT, nullptr, &ipw, wArA)) return -1;
            for (int i = 1; i < nc; i++) wArA[i] = wArA[0];
        }
        else if (cred<CR_DOT>(m, W, wA, bi ? rT : rA, nullptr, nullptr, wArA)) return -1;

        if (perf->nIterations == 0)
        {
            if (cew<CEW_COPY>(m, W, pA, wA, nullptr)) return -1;
            if (bi && cew<CEW_COPY>(m, W, pT, wT, nullptr)) return -1;
        }
        else
        {
            Cmpts beta{};
            for (int i = 0; i < nc; i++)
                beta.v[i] = bi == 2 ? wArA[i] / wArAold[i] : wArA[i] / stabilise(wArAold[i], kVSmall);
            if (cew<CEW_P_UPDATE>(m, W, pA, wA, nullptr, &beta)) return -1;
            if (bi && cew<CEW_P_UPDATE>(m, W, pT, wT, nullptr, &beta)) return -1;
        }
        if (c_amul(m, W, wA, pA, false)) return -1;
        if (bi && c_amul(m, W, wT, pT, true)) return -1;
        if (bi == 2)
        {
            if (cred<CR_DOTALL>(m, W, wA, pT, nullptr, &ipw, wApA)) return -1;
            for (int i = 1; i < nc; i++) wApA[i] = wApA[0];
        }
        else if (cred<CR_DOT>(m, W, wA, bi ? pT : pA, nullptr, nullptr, wApA)) return -1;

        double test[LDU_MAX_CMPT];
        for (int i = 0; i < nc; i++) test[i] = std::fabs(wApA[i]) / nf[i];
        if (check_singularity(perf, nc, test)) break;

        Cmpts alpha{};
        for (int i = 0; i < nc; i++)
            alpha.v[i] = bi == 2 ? wArA[i] / wApA[i] : wArA[i] / stabilise(wApA[i], kVSmall);
        if (cew<CEW_ADD_K>(m, W, psi, pA, nullptr, &alpha)) return -1;
        if (cew<CEW_SUB_K>(m, W, rA, wA, nullptr, &alpha)) return -1;
        if (bi && cew<CEW_SUB_K>(m, W, rT, wT, nullptr, &alpha)) return -1;
        if (cred<CR_SUMMAG>(m, W, rA, nullptr, nullptr, nullptr, res)) return -1;
        for (int i = 0; i < nc; i++) perf->finalResidual[i] = res[i] / nf[i];

        if (!(perf->nIterations++ < c->maxIter)) break;
        if (check_convergence(perf, c)) break;
    }
    return 0;
}

// SmoothSolver.C:61-151
static int solve_smooth(ldu_matrix* m, CoupledWork* W, const ldu_coupled_controls* c, double* psi,
                        const double* source, ldu_coupled_perf* perf)
{
    const int nc = W->nc;
    if (c->nSweeps < 0)
    {
        if (c_smooth(m, W, psi, source, -c->nSweeps)) return -1;
        perf->nIterations -= c->nSweeps;
        return 0;
    }
    double* Apsi = field(W, 3);
    double* rA = field(W, 4);
    if (!Apsi || !rA) return -1;
    double res[LDU_MAX_CMPT];
    double* nf = perf->normFactor;
    if (c_amul(m, W, Apsi, psi, false)) return -1;
    if (norm_factor(m, W, psi, source, Apsi, nf)) return -1;
    if (cew<CEW_SUB>(m, W, rA, source, Apsi)) return -1;
    if (cred<CR_SUMMAG>(m, W, rA, nullptr, nullptr, nullptr, res)) return -1;
    for (int i = 0; i < nc; i++) perf->initialResidual[i] = perf->finalResidual[i] = res[i] / nf[i];
    if (check_convergence(perf, c)) return 0;
    for (;;)
    {
        if (c_smooth(m, W, psi, source, c->nSweeps)) return -1;
        if (c_residual(m, W, rA, psi, source)) return -1;
        if (cred<CR_SUMMAG>(m, W, rA, nullptr, nullptr, nullptr, res)) return -1;
        for (int i = 0; i < nc; i++) perf->finalResidual[i] = res[i] / nf[i];
        if (!((perf->nIterations += c->nSweeps) < c->maxIter)) break;
        if (check_convergence(perf, c)) break;
    }
    return 0;
}

// ---------------------------------------------------------------- C ABI
static int need(ldu_matrix* m, int nc)
{
    if (!m || !m->haveCoeffs) { ldu_set_error("ldu_matrix_set_coeffs() not called"); return -15; }
    if (nc < 1 || nc > LDU_MAX_CMPT) { ldu_set_error("nCmpt must be 1..9"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(m->a->ctx->device));
    return 0;
}

extern "C" {

void ldu_coupled_default_controls(ldu_coupled_controls* c, int32_t nCmpt)
{
    memset(c, 0, sizeof(*c));
    c->solver = LDU_CSOLVER_PBICCCG;
    c->preconditioner = LDU_CPRE_DILU;
    c->smoother = LDU_CSM_GAUSSSEIDEL;
    c->nCmpt = nCmpt;
    c->maxIter = 1000;    // LduMatrixSolver.C:134-136
    c->nSweeps = 1;       // SmoothSolver.C:41
    for (int i = 0; i < LDU_MAX_CMPT; i++) { c->tolerance[i] = 1e-6; c->relTol[i] = 0.0; c->innerProductWeights[i] = 1.0; }
    if (nCmpt == 6) { c->innerProductWeights[1] = c->innerProductWeights[2] = c->innerProductWeights[4] = 2.0; }
}

int ldu_coupled_solve(ldu_matrix* m, const ldu_coupled_controls* c, double* psi, const double* source,
                      ldu_coupled_perf* perf)
{
    if (int rc = need(m, c->nCmpt)) return rc;
    memset(perf, 0, sizeof(*perf));
    ldu_addr* a = m->a;
    // LduMatrixSolver.C:45-110: the matrix decides the table, a name outside it is fatal
    int solver = c->solver;
    const bool diagonal = a->nFaces == 0;
    if (diagonal) solver = LDU_CSOLVER_DIAGONAL;
    else if (m->sym && (solver == LDU_CSOLVER_PBICCCG || solver == LDU_CSOLVER_PBICICG))
    {
        ldu_set_error("Unknown symmetric matrix solver: PBiCCCG / PBiCICG are asymmetric-matrix solvers (lduSolvers.C:41-45)");
        return -16;
    }
    else if (!m->sym && solver == LDU_CSOLVER_PCICG)
    {
        ldu_set_error("Unknown asymmetric matrix solver PCICG (lduSolvers.C:38-39)");
        return -16;
    }
    if (solver < 0 || solver > LDU_CSOLVER_DIAGONAL) { ldu_set_error("unknown coupled solver"); return -3; }
    if (solver != LDU_CSOLVER_DIAGONAL && solver != LDU_CSOLVER_SMOOTHSOLVER)
    {
        if (m->sym && c->preconditioner == LDU_CPRE_DILU)
        {
            ldu_set_error("Unknown symmetric matrix preconditioner DILU (lduPreconditioners.C:41-42)");
            return -16;
        }
        if (c->preconditioner < 0 || c->preconditioner > LDU_CPRE_DILU)
        {
            ldu_set_error("unknown coupled preconditioner");
            return -3;
        }
    }
    if (solver == LDU_CSOLVER_SMOOTHSOLVER && c->smoother != LDU_CSM_GAUSSSEIDEL)
    {
        ldu_set_error("unknown coupled smoother");
        return -3;
    }
    return run_with_fallback(m, [&]() -> int {
    memset(perf, 0, sizeof(*perf));
    CoupledWork* W = work_of(m, c->nCmpt);
    if (!W) { ldu_set_error("coupled work allocation failed"); return -1; }
    double* x = field_in(m, W, 0, 0, psi);
    double* b = field_in(m, W, 1, 1, source);
    if (!x || !b) { ldu_set_error("coupled field staging failed"); return -1; }
    hipStream_t s = a->ctx->stream;
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0;
    switch (solver)
    {
    case LDU_CSOLVER_DIAGONAL:   // DiagonalSolver.C:56-76: psi = source/diag, zero residuals, converged
        rc = cew<CEW_DIV_S>(m, W, x, b, m->d_diag);
        perf->converged = 1;
        break;
    case LDU_CSOLVER_PCICG: rc = solve_krylov(m, W, c, x, b, perf, 0); break;
    case LDU_CSOLVER_PBICICG: rc = solve_krylov(m, W, c, x, b, perf, 1); break;
    case LDU_CSOLVER_PBICCCG: rc = solve_krylov(m, W, c, x, b, perf, 2); break;
    case LDU_CSOLVER_SMOOTHSOLVER: rc = solve_smooth(m, W, c, x, b, perf); break;
    }
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    perf->solveSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    if (int rc2 = dev_check_abort(a->ctx)) return rc2;
    return field_out(m, W, psi, x);
    });
}

int ldu_coupled_amul(ldu_matrix* m, int32_t nCmpt, double* Apsi, const double* psi, int32_t transpose)
{
    if (int rc = need(m, nCmpt)) return rc;
    CoupledWork* W = work_of(m, nCmpt);
    if (!W) return -1;
    double* x = field_in(m, W, 0, 0, psi);
    double* y = field(W, 3);
    if (!x || !y) return -1;
    if (c_amul(m, W, y, x, transpose != 0)) return -1;
    return field_out(m, W, Apsi, y);
}

int ldu_coupled_residual(ldu_matrix* m, int32_t nCmpt, double* rA, const double* psi, const double* source)
{
    if (int rc = need(m, nCmpt)) return rc;
    CoupledWork* W = work_of(m, nCmpt);
    if (!W) return -1;
    double* x = field_in(m, W, 0, 0, psi);
    double* b = field_in(m, W, 1, 1, source);
    double* r = field(W, 4);
    if (!x || !b || !r) return -1;
    if (c_residual(m, W, r, x, b)) return -1;
    return field_out(m, W, rA, r);
}

int ldu_coupled_precondition(ldu_matrix* m, int32_t pre, int32_t nCmpt, double* wA, const double* rA,
                             int32_t transpose)
{
    if (int rc = need(m, nCmpt)) return rc;
    if (m->sym && pre == LDU_CPRE_DILU)
    {
        ldu_set_error("Unknown symmetric matrix preconditioner DILU (lduPreconditioners.C:41-42)");
        return -16;
    }
    return run_with_fallback(m, [&]() -> int {
        CoupledWork* W = work_of(m, nCmpt);
        if (!W) return -1;
        double* r = field_in(m, W, 1, 1, rA);
        double* w = field(W, 3);
        if (!r || !w) return -1;
        if (c_precondition(m, W, pre, w, r, transpose != 0)) return -1;
        if (int rc = dev_check_abort(m->a->ctx)) return rc;
        return field_out(m, W, wA, w);
    });
}

int ldu_coupled_smooth(ldu_matrix* m, int32_t smoother, int32_t nCmpt, double* psi, const double* source,
                       int32_t nSweeps)
{
    if (int rc = need(m, nCmpt)) return rc;
    if (smoother != LDU_CSM_GAUSSSEIDEL) { ldu_set_error("unknown coupled smoother"); return -3; }
    if (m->a->nFaces == 0)
    {
        ldu_set_error("cannot solve incomplete matrix, no off-diagonal coefficients (LduMatrixSmoother.C:96-104)");
        return -16;
    }
    return run_with_fallback(m, [&]() -> int {
        CoupledWork* W = work_of(m, nCmpt);
        if (!W) return -1;
        double* x = field_in(m, W, 0, 0, psi);
        double* b = field_in(m, W, 1, 1, source);
        if (!x || !b) return -1;
        if (c_smooth(m, W, x, b, nSweeps)) return -1;
        if (int rc = dev_check_abort(m->a->ctx)) return rc;
        return field_out(m, W, psi, x);
    });
}

}  // extern "C"
