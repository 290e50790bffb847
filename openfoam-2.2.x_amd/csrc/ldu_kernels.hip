// HIP kernels of the lduMatrix hot path for gfx950 (CDNA4, wave64).
//
// Everything here is HBM-bandwidth- or dependency-latency-bound f64 work with int32
// indirection: no MFMA.  Rules followed (cdna_hip_programming.md section 6):
//   * one wavefront (64 lanes) owns one slice of <= 64 consecutive rows; entry k of the
//     slice is read as 64 consecutive int32 / f64 values (fully coalesced sliced-ELL);
//   * reductions: per-lane accumulation -> wave shuffle tree -> LDS across the 4 waves of a
//     block -> one partial per block -> single-block final pass (deterministic, no atomics);
//   * compiled with -ffp-contract=off: a*b+c is never contracted, so every row reproduces
//     the reference's rounding exactly (SURVEY.md Appendix B last bullet).
#include <cmath>
#include <algorithm>

#include "ldu_peer_dev.hpp"

#define BLK 256
#define WPB (BLK / LDU_WAVE)   // waves per block

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- layout kernels

__global__ void fill_sell_kernel(int nSlices, const int* __restrict__ sliceRow,
                                 const int* __restrict__ sliceCnt, const int* __restrict__ sliceEnt,
                                 const int* __restrict__ sliceW, const unsigned char* __restrict__ nL,
                                 const int* __restrict__ face, const double* __restrict__ lowerO,
                                 const double* __restrict__ upperO, double* __restrict__ val)
{
    const int s = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (s >= nSlices) return;
    const int lane = threadIdx.x & 63;
    const int cnt = sliceCnt[s];
    const int r = sliceRow[s] + lane;
    const int nl = lane < cnt ? nL[r] : 0;
    const long ent = (long)sliceEnt[s] + lane;
    const int W = sliceW[s];
    for (int k0 = 0; k0 < W; k0 += 8)
    {
        // eight face indices, then eight coefficient gathers in flight
        int f[8];
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (k0 + j < W) f[j] = face[ent + (long)(k0 + j) * LDU_WAVE];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (k0 + j < W)
            {
                v[j] = 0.0;
                if (f[j] >= 0) v[j] = (k0 + j < nl) ? lowerO[f[j]] : upperO[f[j]];
            }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (k0 + j < W) val[ent + (long)(k0 + j) * LDU_WAVE] = v[j];
    }
}

int k_fill_sell(ldu_addr* a, const double* lowerO, const double* upperO, double* val, hipStream_t s)
{
    if (a->nSlices == 0) return 0;
    val_touch(a, val);
    fill_sell_kernel<<<cdiv(a->nSlices, WPB), BLK, 0, s>>>(a->nSlices, a->d_sliceRow, a->d_sliceCnt,
        a->d_sliceEnt, a->d_sliceW, a->d_nL, a->d_face, lowerO, upperO, val);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// the coefficient array of a per-sweep layout (ldu_gslayouts.cpp): the same fill, the layout's own tables (nL per slot)
int k_fill_layout(const ldu_addr::GsLayout* Y, const double* lowerO, const double* upperO, double* val, hipStream_t s)
{
    if (Y->nSlices == 0) return 0;
    fill_sell_kernel<<<cdiv(Y->nSlices, WPB), BLK, 0, s>>>(Y->nSlices, Y->d_sliceRow, Y->d_sliceCnt, Y->d_sliceEnt, Y->d_sliceW,
                                                           Y->d_nL, Y->d_face, lowerO, upperO, val);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ void scale_rows_kernel(int nSlices, const int* __restrict__ sliceRow,
                                  const int* __restrict__ sliceCnt, const int* __restrict__ sliceEnt,
                                  const int* __restrict__ sliceW, const double* __restrict__ valIn,
                                  const double* __restrict__ rowScale, double* __restrict__ valOut)
{
    const int s = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (s >= nSlices) return;
    const int lane = threadIdx.x & 63;
    const int cnt = sliceCnt[s];
    const int r = sliceRow[s] + lane;
    const double sc = lane < cnt ? rowScale[r] : 0.0;
    const long ent = (long)sliceEnt[s] + lane;
    const int W = sliceW[s];
    for (int k = 0; k < W; k++)
    {
        const long e = ent + (long)k * LDU_WAVE;
        valOut[e] = sc * valIn[e];
    }
}

int k_scale_rows(ldu_addr* a, double* valOut, const double* valIn, const double* rowScale, hipStream_t s)
{
    if (a->nSlices == 0) return 0;
    val_touch(a, valOut);
    scale_rows_kernel<<<cdiv(a->nSlices, WPB), BLK, 0, s>>>(a->nSlices, a->d_sliceRow, a->d_sliceCnt,
        a->d_sliceEnt, a->d_sliceW, valIn, rowScale, valOut);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ void permute_in_kernel(int n, const int* __restrict__ perm, const double* __restrict__ src,
                                  double* __restrict__ dst)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) dst[i] = src[perm[i]];
}
__global__ void permute_out_kernel(int n, const int* __restrict__ perm, const double* __restrict__ src,
                                   double* __restrict__ dst)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) dst[perm[i]] = src[i];
}

static inline int ewGrid(long n) { int g = cdiv(n, BLK); return g < 1 ? 1 : (g > 4096 ? 4096 : g); }

int k_permute_in(ldu_addr* a, double* dstNew, const double* srcOld, hipStream_t s)
{
    permute_in_kernel<<<ewGrid(a->nCells), BLK, 0, s>>>(a->nCells, a->d_perm, srcOld, dstNew);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}
int k_permute_out(ldu_addr* a, double* dstOld, const double* srcNew, hipStream_t s)
{
    permute_out_kernel<<<ewGrid(a->nCells), BLK, 0, s>>>(a->nCells, a->d_perm, srcNew, dstOld);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ void reciprocal_kernel(int n, double* x)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) x[i] = 1.0 / x[i];
}
int k_reciprocal(int n, double* x, hipStream_t s)
{
    reciprocal_kernel<<<ewGrid(n), BLK, 0, s>>>(n, x);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- row kernels (Amul family)
// MODE 0: y = diag*x + sum val*x[col]                 (Amul / Tmul, lduMatrixATmul.C:34-151)
// MODE 1: y = b - diag*x - sum val*x[col]             (residual, :203-280)
// MODE 2: y = diag + sum val                          (sumA, :154-200)
// MODE 3: y = - sum val*x[col]                        (H, lduMatrixTemplates.C:34-65)
// MODE 4: y = - sum val                               (H1, lduMatrixATmul.C:298-327)
// MODE 5: y = + sum val*x[col]                        (GAMG interpolate, GAMGSolverInterpolate.C:61-66)
template <int MODE>
__global__ void __launch_bounds__(BLK)
row_kernel(int nSlices, const int* __restrict__ sliceRow, const int* __restrict__ sliceCnt,
           const int* __restrict__ sliceEnt, const int* __restrict__ sliceW,
           const unsigned char* __restrict__ nL, const unsigned char* __restrict__ nU,
           const int* __restrict__ col, const double* __restrict__ val,
           const double* __restrict__ diag, const double* __restrict__ x,
           const double* __restrict__ b, double* __restrict__ y)
{
    const int s = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (s >= nSlices) return;
    const int lane = threadIdx.x & 63;
    const int cnt = sliceCnt[s];
    if (lane >= cnt) return;
    const int r = sliceRow[s] + lane;
    const int n = (int)nL[r] + (int)nU[r];
    const long ent = (long)sliceEnt[s] + lane;
    double acc;
    if (MODE == 0) acc = diag[r] * x[r];
    else if (MODE == 1) acc = b[r] - diag[r] * x[r];
    else if (MODE == 2) acc = diag[r];
    else acc = 0.0;
    const int W = sliceW[s];   // wave-uniform
    if (MODE == 0 || MODE == 1 || MODE == 3 || MODE == 5)
    {
        // eight entries at a time: their columns and coefficients, then their x values, in flight together (one dependent
        // gather per trip - the loop below - cost the agglomerated levels of an unstructured mesh, whose slices are 9-33 entries
        // wide, as much time for half the rows as the finest level); the accumulation order and the k < n guard are unchanged
        for (int k0 = 0; k0 < W; k0 += 8)
        {
            int c[8];
            double v[8], xv[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k0 + k < W)
                {
                    const long e = ent + (long)(k0 + k) * LDU_WAVE;
                    // columns and coefficients are streamed once: nontemporal, so that L2 keeps the x values the gathers
                    // come back for (216^3 Amul 0.1838 -> 0.1784 ms); padding entries point at the row itself
                    c[k] = __builtin_nontemporal_load(col + e);
                    v[k] = __builtin_nontemporal_load(val + e);
                }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k0 + k < W) xv[k] = x[c[k]];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k0 + k < W && k0 + k < n)
                {
                    if (MODE == 0 || MODE == 5) acc += v[k] * xv[k];
                    else acc -= v[k] * xv[k];
                }
        }
        y[r] = acc;
        return;
    }
    for (int k = 0; k < n; k++)
    {
        const long e = ent + (long)k * LDU_WAVE;
        const double v = val[e];
        if (MODE == 0 || MODE == 5) acc += v * x[col[e]];
        else if (MODE == 1 || MODE == 3) acc -= v * x[col[e]];
        else if (MODE == 2) acc += v;
        else acc -= v;
    }
    y[r] = acc;
}

template <int MODE>
static int launch_row(ldu_matrix* m, double* y, const double* x, const double* b, const double* val,
                      hipStream_t s)
{
    ldu_addr* a = m->a;
    if (a->nSlices == 0) return 0;
    row_kernel<MODE><<<cdiv(a->nSlices, WPB), BLK, 0, s>>>(a->nSlices, a->d_sliceRow, a->d_sliceCnt,
        a->d_sliceEnt, a->d_sliceW, a->d_nL, a->d_nU, a->d_col, val, m->d_diag, x, b, y);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

int k_amul(ldu_matrix* m, double* y, const double* x, bool transpose, hipStream_t s)
{
    m->a->ctx->profStart(m->a, LDU_PROF_AMUL);
    int rc = launch_row<0>(m, y, x, nullptr, transpose ? m->d_valT : m->d_valA, s);
    m->a->ctx->profStop(m->a, LDU_PROF_AMUL);
    return rc;
}
int k_residual_rows(ldu_matrix* m, double* r, const double* x, const double* b, hipStream_t s)
{
    m->a->ctx->profStart(m->a, LDU_PROF_RESIDUAL);
    int rc = launch_row<1>(m, r, x, b, m->d_valA, s);
    m->a->ctx->profStop(m->a, LDU_PROF_RESIDUAL);
    return rc;
}
int k_sumA_rows(ldu_matrix* m, double* sumA, hipStream_t s)
{
    return launch_row<2>(m, sumA, nullptr, nullptr, m->d_valA, s);
}
int k_offdiag(ldu_matrix* m, double* y, const double* x, int mode, hipStream_t s)
{
    if (mode == 0) return launch_row<3>(m, y, x, nullptr, m->d_valA, s);
    if (mode == 1) return launch_row<4>(m, y, nullptr, nullptr, m->d_valA, s);
    return launch_row<5>(m, y, x, nullptr, m->d_valA, s);
}

// faceH (lduMatrixTemplates.C:78-110) on the original numbering: x in ORIGINAL order
__global__ void faceH_kernel(int nFaces, const int* __restrict__ l, const int* __restrict__ u,
                             const double* __restrict__ lower, const double* __restrict__ upper,
                             const double* __restrict__ x, double* __restrict__ out)
{
    for (int f = blockIdx.x * BLK + threadIdx.x; f < nFaces; f += gridDim.x * BLK)
        out[f] = upper[f] * x[u[f]] - lower[f] * x[l[f]];
}
int k_faceH(ldu_matrix* m, double* faceH, const double* xOld, hipStream_t s)
{
    ldu_addr* a = m->a;
    if (a->nFaces == 0) return 0;
    faceH_kernel<<<ewGrid(a->nFaces), BLK, 0, s>>>(a->nFaces, a->d_l, a->d_u, m->d_lowerO, m->d_upperO,
                                                   xOld, faceH);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- level-scheduled sweeps

struct SliceTab {
    const int* sliceRow; const int* sliceCnt; const int* sliceEnt;
    const unsigned char* nL; const unsigned char* nU; const int* col;
    const int* sliceW = nullptr;      // per slice: entries per row (max nL+nU)
    const unsigned char* sliceT = nullptr;   // per slice: lanes per row (cooperative slices: 2 / 4 / 8), or no table
    const int* gate = nullptr;        // per slice: slice whose completion opens the polling gate (-1: none)
    unsigned* sliceDone = nullptr;    // per slice: tag of the last sweep that completed it (hint only)
    const int* rowIdx = nullptr;      // per-sweep layouts (ldu_gslayouts.cpp): slot -> row of the level numbering; nL / nU are per slot
};

template <int MODE>
__device__ __forceinline__ void sweep_slice(const SliceTab& T, int s, int lane, double* __restrict__ w,
                                            const double* __restrict__ rhs,
                                            const double* __restrict__ scale,
                                            const double* __restrict__ val,
                                            const double* __restrict__ val2, double* __restrict__ aux)
{
    const int cnt = T.sliceCnt[s];
    if (lane >= cnt) return;
    const int r = T.sliceRow[s] + lane;
    const int nl = T.nL[r];
    const int nu = T.nU[r];
    const long ent = (long)T.sliceEnt[s] + lane;
    constexpr int B = sw_base(MODE);
    constexpr bool TF = sw_tform(MODE);
    if (B == SW_TRI_FWD)
    {
        // DICPreconditioner.C:109-117 / DILUPreconditioner.C:113-128 with valP = rD[row]*coeff
        const double sc = scale[r];
        double acc = sc * rhs[r];
        for (int k = 0; k < nl; k++)
        {
            const long e = ent + (long)k * LDU_WAVE;
            if (TF) acc -= sc * (val[e] * w[T.col[e]]);
            else acc -= val[e] * w[T.col[e]];
        }
        w[r] = acc;
    }
    else if (B == SW_TRI_BWD)
    {
        // DICPreconditioner.C:119-122: owned faces in DEscending order
        double acc = w[r];
        const double sc = TF ? scale[r] : 0.0;
        for (int k = nl + nu - 1; k >= nl; k--)
        {
            const long e = ent + (long)k * LDU_WAVE;
            if (TF) acc -= sc * (val[e] * w[T.col[e]]);
            else acc -= val[e] * w[T.col[e]];
        }
        w[r] = acc;
    }
    else if (B == SW_RD)
    {
        // DICPreconditioner.C:71-74 / DILUPreconditioner.C:72-75 (before the reciprocal)
        double acc = scale[r];
        for (int k = 0; k < nl; k++)
        {
            const long e = ent + (long)k * LDU_WAVE;
            if (TF) acc -= (val2[e] * val[e]) * (1.0 / w[T.col[e]]);
            else acc -= (val2[e] * val[e]) / w[T.col[e]];
        }
        w[r] = acc;
    }
    else if (B == SW_GS_FWD)
    {
        // GaussSeidelSmoother.C:151-176 as a row gather
        double acc = rhs[r];
        for (int k = 0; k < nl; k++)
        {
            const long e = ent + (long)k * LDU_WAVE;
            acc -= val[e] * w[T.col[e]];
        }
        if (aux) aux[r] = acc;
        for (int k = nl; k < nl + nu; k++)
        {
            const long e = ent + (long)k * LDU_WAVE;
            acc -= val[e] * w[T.col[e]];
        }
        w[r] = TF ? scale[r] * acc : acc / scale[r];
    }
    else   // SW_GS_BWD: symGaussSeidelSmoother.C:178-205
    {
        double acc = rhs[r];
        for (int k = nl; k < nl + nu; k++)
        {
            const long e = ent + (long)k * LDU_WAVE;
            acc -= val[e] * w[T.col[e]];
        }
        w[r] = acc / scale[r];
    }
}

template <int MODE>
__global__ void __launch_bounds__(BLK)
sweep_level_kernel(SliceTab T, int sliceBegin, int sliceEnd, double* w, const double* rhs,
                   const double* scale, const double* val, const double* val2, double* aux)
{
    const int s = sliceBegin + blockIdx.x * WPB + (threadIdx.x >> 6);
    if (s >= sliceEnd) return;
    sweep_slice<MODE>(T, s, threadIdx.x & 63, w, rhs, scale, val, val2, aux);
}

#define FUSED_THREADS 1024
// One block walks a run of small levels; __syncthreads() orders the levels (same-CU
// visibility of global stores is all that is needed).
template <int MODE, bool DESC>
__global__ void __launch_bounds__(FUSED_THREADS)
sweep_fused_kernel(SliceTab T, const int* __restrict__ levelSliceStart, int levelBegin, int levelEnd,
                   double* w, const double* rhs, const double* scale, const double* val,
                   const double* val2, double* aux)
{
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int nW = FUSED_THREADS / LDU_WAVE;
    if (!DESC)
    {
        for (int L = levelBegin; L < levelEnd; L++)
        {
            const int s1 = levelSliceStart[L + 1];
            for (int s = levelSliceStart[L] + wave; s < s1; s += nW)
                sweep_slice<MODE>(T, s, lane, w, rhs, scale, val, val2, aux);
            __syncthreads();
        }
    }
    else
    {
        for (int L = levelEnd - 1; L >= levelBegin; L--)
        {
            const int s1 = levelSliceStart[L + 1];
            for (int s = levelSliceStart[L] + wave; s < s1; s += nW)
                sweep_slice<MODE>(T, s, lane, w, rhs, scale, val, val2, aux);
            __syncthreads();
        }
    }
}

template <int MODE, bool DESC>
static int launch_sweep_segments(ldu_addr* a, const SweepArgs& g, hipStream_t s)
{
    SliceTab T{a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_nL, a->d_nU, a->d_col};
    const int nSeg = (int)a->segs.size();
    for (int i = 0; i < nSeg; i++)
    {
        const Segment& sg = a->segs[DESC ? nSeg - 1 - i : i];
        if (sg.sliceEnd == sg.sliceBegin) continue;
        if (sg.fused)
            sweep_fused_kernel<MODE, DESC><<<1, FUSED_THREADS, 0, s>>>(T, a->d_levelSliceStart,
                sg.levelBegin, sg.levelEnd, g.w, g.rhs, g.scale, g.val, g.val2, g.aux);
        else
            sweep_level_kernel<MODE><<<cdiv(sg.sliceEnd - sg.sliceBegin, WPB), BLK, 0, s>>>(
                T, sg.sliceBegin, sg.sliceEnd, g.w, g.rhs, g.scale, g.val, g.val2, g.aux);
    }
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

static int launch_sweep(ldu_addr* a, const SweepArgs& g, hipStream_t s)
{
    switch (g.mode)
    {
    case SW_TRI_FWD: return launch_sweep_segments<SW_TRI_FWD, false>(a, g, s);
    case SW_TRI_BWD: return launch_sweep_segments<SW_TRI_BWD, true>(a, g, s);
    case SW_RD:      return launch_sweep_segments<SW_RD, false>(a, g, s);
    case SW_GS_FWD:  return launch_sweep_segments<SW_GS_FWD, false>(a, g, s);
    case SW_GS_BWD:  return launch_sweep_segments<SW_GS_BWD, true>(a, g, s);
    case SW_TRI_FWD_T: return launch_sweep_segments<SW_TRI_FWD_T, false>(a, g, s);
    case SW_TRI_BWD_T: return launch_sweep_segments<SW_TRI_BWD_T, true>(a, g, s);
    case SW_RD_T:      return launch_sweep_segments<SW_RD_T, false>(a, g, s);
    case SW_GS_FWD_T:  return launch_sweep_segments<SW_GS_FWD_T, false>(a, g, s);
    }
    return -1;
}

// nonBlockingGaussSeidelSmoother.C:147-237 as a row gather, one launch per dependency level.  Same
// values as GaussSeidel; a row with coupled faces adds their contributions AFTER its lower neighbours
// below blockStart and BEFORE the others (the reference sweeps the cells below blockStart before the
// halo arrives).  Rarely selected, so it runs on the simple level kernels, not on the sweep engines.
__global__ void __launch_bounds__(BLK)
gs_nonblocking_level_kernel(SliceTab T, int sliceBegin, int sliceEnd, const int* __restrict__ rowB,
                            const unsigned char* __restrict__ k0s, const int* __restrict__ bStart,
                            const int* __restrict__ bFace, const double* __restrict__ bou,
                            const double* __restrict__ recv, double* __restrict__ psi,
                            const double* __restrict__ source, const double* __restrict__ diag,
                            const double* __restrict__ val)
{
    const int s = sliceBegin + blockIdx.x * WPB + (threadIdx.x >> 6);
    if (s >= sliceEnd) return;
    const int lane = threadIdx.x & 63;
    if (lane >= T.sliceCnt[s]) return;
    const int r = T.sliceRow[s] + lane;
    const int nl = T.nL[r], nu = T.nU[r];
    const long ent = (long)T.sliceEnt[s] + lane;
    const int bi = rowB[r];
    const int k0 = bi >= 0 ? (int)k0s[r] : nl;
    double acc = source[r];
    for (int k = 0; k < k0; k++)
    {
        const long e = ent + (long)k * LDU_WAVE;
        acc -= val[e] * psi[T.col[e]];
    }
    if (bi >= 0)
    {
        // updateMatrixInterfaces with the negated coefficients (:120-128, :198-205):
        // bPrime[faceCells] -= (-bouCoeffs)*pnf, patch by patch, face by face
        for (int j = bStart[bi]; j < bStart[bi + 1]; j++)
        {
            const int f = bFace[j];
            acc -= (-bou[f]) * recv[f];
        }
        for (int k = k0; k < nl; k++)
        {
            const long e = ent + (long)k * LDU_WAVE;
            acc -= val[e] * psi[T.col[e]];
        }
    }
    for (int k = nl; k < nl + nu; k++)
    {
        const long e = ent + (long)k * LDU_WAVE;
        acc -= val[e] * psi[T.col[e]];
    }
    psi[r] = acc / diag[r];
}

// one sweep; the halo (recv buffers) must have been exchanged by the caller
int k_sweep_gs_nonblocking(ldu_addr* a, double* psi, const double* source, const double* diag, const double* val,
                           const double* bou)
{
    SliceTab T{a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_nL, a->d_nU, a->d_col};
    hipStream_t s = a->ctx->stream;
    if (comm_wait_halo(a, s)) return -1;
    for (int L = 0; L < a->nLevels; L++)
    {
        const int s0 = a->levelSliceStart[L], s1 = a->levelSliceStart[L + 1];
        if (s1 == s0) continue;
        gs_nonblocking_level_kernel<<<cdiv(s1 - s0, WPB), BLK, 0, s>>>(T, s0, s1, a->d_nbRowB, a->d_nbK0,
            a->d_bStart, a->d_bFace, bou, a->d_recvAll, psi, source, diag, val);
    }
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- persistent point-to-point sweep
//
// One launch per sweep.  Slices are taken in dependency-level order through a chunk ticket
// (atomicAdd, one per CHUNK slices), so every slice a wave can wait for has already been taken
// by a RUNNING workgroup: forward progress without any co-residency assumption.
// A finished row publishes {value, tag = launch epoch} as two 8-byte agent-scope (sc1,
// write-through) granules {v_lo, tag}, {v_hi, tag}; a consumer polls the 16 bytes with sc1 loads
// (L2/fabric served, never the stale per-CU L1) until both tags match - no flags, no fences, no
// reset pass (MI355X_MICROARCH.md "handoff-1to1", cdna_hip_programming.md G16 form R2).
// Everything a row needs that does not depend on other rows of this sweep (coefficients,
// columns, rhs) is loaded before the wait, so the critical path per dependency level is one
// cross-CU hand-off instead of a kernel boundary plus a chain of dependent HBM loads.
// Spins are bounded: on expiry the sweep sets *abortFlag and every wave drains.

#ifndef P2P_BLK
#define P2P_BLK 256
#endif
#define P2P_CHUNK (P2P_BLK / LDU_WAVE)   // slices per ticket: one per wave of the workgroup
#define P2P_SPIN_LIMIT_DEFAULT (1u << 22)
// bound of every dependency wait, in polls; run-time overridable (ldu_ctx_set_spin_limit / LDU_SPIN_LIMIT) so that
// the abort -> engine-fallback path can be forced in tests
__device__ unsigned g_p2p_spin_limit = P2P_SPIN_LIMIT_DEFAULT;
int k_set_spin_limit(unsigned polls)
{
    if (!polls) polls = P2P_SPIN_LIMIT_DEFAULT;
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_spin_limit), &polls, sizeof(unsigned)));
    return 0;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// optional per-slice trace (LDU_P2P_TRACE): {tTicket, tWait, tReady, tDone, polls, xcc, slice, 0}
struct P2PStat { long long tWait, tReady; unsigned polls; int gateSlice; const unsigned* sliceDone; };
__device__ unsigned long long* g_p2p_trace = nullptr;
int k_set_p2p_trace(unsigned long long* buf)
{
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_trace), &buf, sizeof(buf)));
    return 0;
}

// first expired wait of a sweep: {claimed, selfRow, expected tag, col0..3, seen tags of col0..3}
__device__ int g_p2p_dbg[16] = {0};
__device__ int g_p2p_dbgN = 0;
__device__ int g_p2p_dbgRec[64 * 8] = {0};   // {kind, row, expected tag, col, seen tag y, seen tag w, sweep/aux, 0}
int k_read_p2p_dbg_records(int* out /* 1 + 64*8 */)
{
    LDU_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2p_dbgN), sizeof(int)));
    LDU_CHECK_HIP(hipMemcpyFromSymbol(out + 1, HIP_SYMBOL(g_p2p_dbgRec), sizeof(int) * 64 * 8));
    int z[64 * 8] = {0};
    int zero = 0;
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_dbgN), &zero, sizeof(int)));
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_dbgRec), z, sizeof(int) * 64 * 8));
    return 0;
}

int k_read_p2p_dbg(int* out)
{
    LDU_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2p_dbg), sizeof(int) * 16));
    int z[16] = {0};
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_dbg), z, sizeof(int) * 16));
    return 0;
}

// an expired wait leaves a record (only then: nothing on the fast path): which row waited for which column, which tag it
// expected and which it saw - ldu_debug_p2p_records
__device__ __forceinline__ void p2p_dbg_record(int kind, int row, unsigned expected, int col, unsigned seenY, unsigned seenW, int aux)
{
    // lower-neighbour waits (kind 1) in records 0..31, upper-neighbour waits (kind 2) in 32..63
    static __device__ int nKind[2] = {0, 0};
    atomicAdd(&g_p2p_dbgN, 1);
    const int q = atomicAdd(&nKind[kind == 1 ? 0 : 1], 1);
    const int i = (kind == 1 ? 0 : 32) + q;
    if (q < 32)
    {
        int* rec = g_p2p_dbgRec + 8 * i;
        rec[0] = kind; rec[1] = row; rec[2] = (int)expected; rec[3] = col; rec[4] = (int)seenY; rec[5] = (int)seenW; rec[6] = aux; rec[7] = 1;
    }
}

// (A nap that grows with the time a wave has been waiting - 24 / 96 / 512 polls -> 8 / 32 / 127 x 64 clocks - was measured:
//  it ends the polling storm of waves that ran far ahead on deep irregular graphs, but such waves then sleep through the
//  arrival of their data: +90 % on a single sweep of the irregular 100^3 graph, -7 % on the 216^3 box benchmark.  What
//  bounds the storm instead is the run-ahead window below.)

// Run-ahead window.  Tickets are drawn in dependency order, so every resident wave beyond the front holds a task and
// polls for data that will not arrive for a long time.  On wide levels that is a few hundred waves for a few
// microseconds; on DEEP graphs (thousands of narrow levels: unstructured numberings) all ~5000 resident waves sit up to
// hundreds of levels ahead and their 16-byte re-reads slow the front itself down by orders of magnitude (irregular 60^3
// graph, two pipelined sweeps, chip-wide engine: > 1 s instead of ~3 ms - the round-1 "spin-bound aborts" at large skew).
// A workgroup therefore starts chunk t only once `done` (chunks reported complete) has reached t - window: the waves
// inside the window are the ones that can do useful prefetching (~8 dependency levels), the others nap on ONE counter.
// Progress: a chunk never waits for a later chunk; the lowest unfinished chunks are inside every window.
__device__ __forceinline__ void p2p_window_wait(const unsigned* done, unsigned doneBase, int t, int window,
                                                volatile int* abortFlag)
{
    if (t < window) return;
    const int need = t - window;
    unsigned naps = 0;
    unsigned long long tw0 = 0;
    while ((int)(__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - doneBase) < need)
    {
        if ((++naps & 63u) == 0)   // (one address for the whole chip: look rarely)
        {
            if (*abortFlag) return;
            const unsigned long long budget = g_wait_budget[0], now = wall_clock64();
            if (!tw0) tw0 = now;
            else if (budget && now - tw0 > budget) { *abortFlag = 1; return; }
        }
        __builtin_amdgcn_s_sleep(16);
    }
}

// poll back-off between two granule polls, in units of s_sleep(1) (64 clocks); tunable (LDU_P2P_SLEEP)
__device__ int g_p2p_wide = 1;    // 8-wide polls for rows with more than four dependencies (LDU_P2P_WIDE=0: groups of four)
int k_set_p2p_wide(int on)
{
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_wide), &on, sizeof(int)));
    return 0;
}
__device__ int g_p2p_sleep = 2;
// Optional back-off of a waiting wave (LDU_P2P_BACKOFF=<polls>, LDU_P2P_BACKOFF_CAP): after that many failed polls
// the nap doubles every 8 polls up to 2^cap x 64 clocks.  Meant for waves far from the front (the first tasks of a
// slab whose predecessors lie in another slab, the trailing sweeps of a pipelined launch).  Measured on the irregular
// 216^3 graph (seven of eight XCDs waiting for the eighth): no gain (4 sweeps 43.3 ms with it, 40.1 without; one
// sweep 21.7 / 21.2) - the run-ahead window already keeps the pollers few.  Off by default.
__device__ unsigned g_p2p_backoff = 0u;    // polls before a waiting wave backs off (0 = never: the default)
__device__ unsigned g_p2p_backoff_cap = 7u;
__device__ __forceinline__ void p2p_nap(unsigned spins, int sleepN)
{
    const unsigned th = g_p2p_backoff;
    if (th == 0u || spins < th) { for (int q = 0; q < sleepN; q++) __builtin_amdgcn_s_sleep(1); return; }
    unsigned sh = 1u + ((spins - th) >> 3);
    const unsigned cap = g_p2p_backoff_cap;
    if (sh > cap) sh = cap;
    const unsigned units = 1u << sh;
    for (unsigned q = 0; q < units; q += 2u) __builtin_amdgcn_s_sleep(2);
}
int k_set_p2p_backoff(unsigned n)
{
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_backoff), &n, sizeof(unsigned)));
    return 0;
}
int k_set_p2p_backoff_cap(unsigned n)
{
    if (n < 1u) n = 1u;
    if (n > 12u) n = 12u;
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_backoff_cap), &n, sizeof(unsigned)));
    return 0;
}
int k_set_p2p_sleep(int n)
{
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_sleep), &n, sizeof(int)));
    return 0;
}

__device__ __forceinline__ void granule_store(uint4* G, int row, double v, unsigned tag)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    u32x4 d;
    d.x = (unsigned)b; d.y = tag; d.z = (unsigned)(b >> 32); d.w = tag;
    uint4* p = G + row;
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}

// XCD-slab engine (see "XCD slabs" below): the granule stays in the producing XCD's L2 (plain store:
// same-XCD consumers hit it there with their L1-bypassing sc1 loads); rows with a neighbour in
// another slab also publish a write-through copy for the other XCDs.
__device__ __forceinline__ void granule_store_slab(uint4* G, uint4* X, int row, double v, unsigned tag,
                                                   bool exported)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    u32x4 d;
    d.x = (unsigned)b; d.y = tag; d.z = (unsigned)(b >> 32); d.w = tag;
    uint4* p = G + row;
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
    if (exported)
    {
        uint4* q = X + row;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(q), "v"(d) : "memory");
    }
}

// four granule loads in flight, one wait
__device__ __forceinline__ void granule_load4(const uint4* p0, const uint4* p1, const uint4* p2,
                                              const uint4* p3, u32x4& g0, u32x4& g1, u32x4& g2,
                                              u32x4& g3)
{
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}

// eight in flight (rows with 5..8 dependencies: one round trip instead of two)
__device__ __forceinline__ void granule_load8(const uint4* const (&p)[8], u32x4 (&g)[8])
{
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %9, off sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc1\n\t"
        "global_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\t"
        "global_load_dwordx4 %7, %15, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "memory");
}

__device__ __forceinline__ double granule_value(const u32x4& g)
{
    return __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
}

// acc -= sum_{i=0..n-1} val[e(i)] * (value of row col[e(i)] published in THIS sweep), in order;
// entry index k(i) = first + i*step.  OP = 0: acc -= v*x ; OP = 1: acc -= (v2*v)/x  (SW_RD);
// templated-family association: OP = 2: acc -= sc*(v*x) ; OP = 3: acc -= (v2*v)*(1/x)
template <int OP, bool DIAG = false, bool SLAB = false>
__device__ __forceinline__ bool p2p_accumulate(double& acc, const uint4* __restrict__ G,
                                               const uint4* __restrict__ X, unsigned tag,
                                               const int* __restrict__ col,
                                               const double* __restrict__ val,
                                               const double* __restrict__ val2, long ent, int first,
                                               int step, int n, int selfRow, volatile int* abortFlag,
                                               P2PStat& waitEst, double sc = 0.0)
{
    // (An adaptive pre-sleep before the first poll was tried and measured 2-6x SLOWER: the wait
    //  shrinks quickly while the levels grow, so any history-based nap oversleeps at the front.)
    if (DIAG && g_p2p_trace) waitEst.tWait = wall_clock64();
    int iStart = 0;
    if (!DIAG && g_p2p_wide && __any(n > 4))
    {
        // Wide rows (agglomerated levels, polyhedral / irregular meshes).  Group by group, every group of four costs
        // two DEPENDENT round trips (its columns and coefficients, then its granules) that start only after the
        // previous group has arrived: rows with 5-8 dependencies spent 2-3 us per dependency level after their last
        // neighbour had published.  Here the first eight columns / coefficients are loaded before the first poll and
        // the eight granules are polled together: one round trip after the last neighbour, as for narrow rows.
        int c8[8];
        double v8[8], w8[8];
        const uint4* gp8[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
            const bool need = j < n;
            const long e = ent + (long)(first + j * step) * LDU_WAVE;
            c8[j] = need ? col[e] : selfRow;
            v8[j] = need ? val[e] : 0.0;
            w8[j] = ((OP == 1 || OP == 3) && need) ? val2[e] : 0.0;
            gp8[j] = SLAB ? (c8[j] < 0 ? X : G) + (c8[j] & 0x7fffffff) : G + c8[j];
        }
        u32x4 g8[8];
        unsigned spins = 0;
            unsigned long long tw0 = 0;
        const int sleepN = g_p2p_sleep;
        const unsigned spinLimit = g_p2p_spin_limit;
        for (;;)
        {
            granule_load8(gp8, g8);
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j < n) ok &= (g8[j].y == tag) & (g8[j].w == tag);
            if (ok) break;
            if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0))
            {
                if (spins > spinLimit)
                {
                    bool rec = false;
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (!rec && j < n && !((g8[j].y == tag) & (g8[j].w == tag)))
                        {
                            p2p_dbg_record(1, selfRow, tag, c8[j], g8[j].y, g8[j].w, j);
                            rec = true;
                        }
                }
                *abortFlag = 1;
                return false;
            }
            p2p_nap(spins, sleepN);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
            if (j < n)
            {
                const double x = granule_value(g8[j]);
                if (OP == 0) acc -= v8[j] * x;
                else if (OP == 1) acc -= (w8[j] * v8[j]) / x;
                else if (OP == 2) acc -= sc * (v8[j] * x);
                else acc -= (w8[j] * v8[j]) * (1.0 / x);
            }
        }
        iStart = 8;
    }
    for (int i0 = iStart; i0 < n; i0 += 4)
    {
        int c[4];
        double v[4], v2[4];
        const uint4* gp[4];   // SLAB: same-slab columns from G (this XCD's L2), others from X
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const bool need = i0 + j < n;
            const long e = ent + (long)(first + (i0 + j) * step) * LDU_WAVE;
            c[j] = need ? col[e] : selfRow;
            v[j] = need ? val[e] : 0.0;
            v2[j] = ((OP == 1 || OP == 3) && need) ? val2[e] : 0.0;
            if (SLAB) gp[j] = (c[j] < 0 ? X : G) + (c[j] & 0x7fffffff);
        }
        if (DIAG && i0 == 0 && waitEst.gateSlice >= 0)
        {
            // Cheap gate before the expensive granule polling (4 x 16 B x 64 lanes per poll):
            // ONE lane polls ONE word - the completion tag of a slice about one dependency level
            // BEFORE the slices this one really waits for.  It is only a hint that the front is
            // near (never used for correctness), so the wave wakes up ~one level early and the
            // fine-grained polling below runs for ~1 hand-off instead of the whole look-ahead.
            const unsigned* gp = waitEst.sliceDone + waitEst.gateSlice;
            unsigned gspins = 0;
            unsigned long long tw0g = 0;
            const unsigned spinLimit = g_p2p_spin_limit;
            for (;;)
            {
                const unsigned gv = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(gv - tag) >= 0) break;
                if (ldu_wait_expired(gspins, spinLimit, abortFlag, tw0g)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            waitEst.gateSlice = -1;
        }
        u32x4 g0, g1, g2, g3;
        unsigned spins = 0;
            unsigned long long tw0 = 0;
        const int sleepN = g_p2p_sleep;
        const unsigned spinLimit = g_p2p_spin_limit;
        for (;; waitEst.polls += DIAG ? 1u : 0u)
        {
            if (SLAB) granule_load4(gp[0], gp[1], gp[2], gp[3], g0, g1, g2, g3);
            else granule_load4(G + c[0], G + c[1], G + c[2], G + c[3], g0, g1, g2, g3);
            bool ok = true;
            if (i0 + 0 < n) ok &= (g0.y == tag) & (g0.w == tag);
            if (i0 + 1 < n) ok &= (g1.y == tag) & (g1.w == tag);
            if (i0 + 2 < n) ok &= (g2.y == tag) & (g2.w == tag);
            if (i0 + 3 < n) ok &= (g3.y == tag) & (g3.w == tag);
            if (ok) break;
            if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0))
            {
                if (spins > spinLimit)
                {
                    if (i0 + 0 < n && !((g0.y == tag) & (g0.w == tag))) p2p_dbg_record(1, selfRow, tag, c[0], g0.y, g0.w, i0);
                    else if (i0 + 1 < n && !((g1.y == tag) & (g1.w == tag))) p2p_dbg_record(1, selfRow, tag, c[1], g1.y, g1.w, i0 + 1);
                    else if (i0 + 2 < n && !((g2.y == tag) & (g2.w == tag))) p2p_dbg_record(1, selfRow, tag, c[2], g2.y, g2.w, i0 + 2);
                    else if (i0 + 3 < n) p2p_dbg_record(1, selfRow, tag, c[3], g3.y, g3.w, i0 + 3);
                }
                *abortFlag = 1;
                return false;
            }
            p2p_nap(spins, sleepN);
        }
        const double x0 = granule_value(g0), x1 = granule_value(g1), x2 = granule_value(g2),
                     x3 = granule_value(g3);
        if (OP == 0)
        {
            if (i0 + 0 < n) acc -= v[0] * x0;
            if (i0 + 1 < n) acc -= v[1] * x1;
            if (i0 + 2 < n) acc -= v[2] * x2;
            if (i0 + 3 < n) acc -= v[3] * x3;
        }
        else if (OP == 1)
        {
            if (i0 + 0 < n) acc -= (v2[0] * v[0]) / x0;
            if (i0 + 1 < n) acc -= (v2[1] * v[1]) / x1;
            if (i0 + 2 < n) acc -= (v2[2] * v[2]) / x2;
            if (i0 + 3 < n) acc -= (v2[3] * v[3]) / x3;
        }
        else if (OP == 2)
        {
            if (i0 + 0 < n) acc -= sc * (v[0] * x0);
            if (i0 + 1 < n) acc -= sc * (v[1] * x1);
            if (i0 + 2 < n) acc -= sc * (v[2] * x2);
            if (i0 + 3 < n) acc -= sc * (v[3] * x3);
        }
        else
        {
            if (i0 + 0 < n) acc -= (v2[0] * v[0]) * (1.0 / x0);
            if (i0 + 1 < n) acc -= (v2[1] * v[1]) * (1.0 / x1);
            if (i0 + 2 < n) acc -= (v2[2] * v[2]) * (1.0 / x2);
            if (i0 + 3 < n) acc -= (v2[3] * v[3]) * (1.0 / x3);
        }
    }
    if (DIAG && g_p2p_trace) waitEst.tReady = wall_clock64();
    return true;
}


// ---------------------------------------------------------------- cooperative rows
// Rows with more than eight lower or upper neighbours (hanging faces of an octree mesh: up to 24; agglomerated GAMG
// levels of such a mesh: 25 ... 66 entries per row) were the critical path of every dependency level they sit in: one
// lane walked the row in groups of four or eight entries, every group two DEPENDENT round trips (columns /
// coefficients, then the granules): 8 - 13 us per level on the octree twin of the motorBike mesh, where a narrow row
// takes 1.5.  Here Tl = 2 / 4 / 8 lanes share a row (the plan puts such rows into slices of 64 / Tl rows of one width
// class): every lane loads, polls and multiplies EIGHT entries of the row's sequence - one round trip for the whole
// row - and parks the products in LDS; the row's first lane then subtracts them in the reference's order (lower part
// ascending, then the upper part; descending for the backward triangular sweep), which is the only part that has to be
// sequential: ~11 ns per entry.  Same products, same order of subtraction: bit-identical.
// Sequence of a row: positions 0 .. nd-1 = the entries whose values THIS sweep produces (polled with `tag`), then, for
// the forward GaussSeidel sweep, nOld = nU positions of "old" upper values (plain loads of w, or the previous
// pipelined sweep's granules, tag - 1).  lds: 512 doubles of this wave.
template <int MODE, bool SLAB, bool MULTI>
__device__ __forceinline__ bool coop_rows(const SliceTab& T, int s, int Tl, int lane, uint4* __restrict__ G,
                                          uint4* __restrict__ X, const unsigned char* __restrict__ xflag, unsigned tag,
                                          bool oldFromGranules, bool writeW, volatile int* abortFlag,
                                          double* __restrict__ w, const double* __restrict__ rhs,
                                          const double* __restrict__ scale, const double* __restrict__ val,
                                          const double* __restrict__ val2, double* __restrict__ aux,
                                          double* __restrict__ lds, unsigned long long* trc = nullptr)
{
    constexpr int B = sw_base(MODE);
    constexpr bool TF = sw_tform(MODE);
    if (trc && lane == 0) trc[0] = (unsigned long long)wall_clock64();
    const int R = LDU_WAVE / Tl;                 // rows of this wave (power of two)
    const int i = lane & (R - 1), t = lane / R;  // row, part: lanes of one part read consecutive rows (coalesced)
    const int cnt = T.sliceCnt[s];
    const bool act = i < cnt;
    const int slot = T.sliceRow[s] + (act ? i : 0);
    const int r = T.rowIdx ? T.rowIdx[slot] : slot;
    const int nl = act ? T.nL[slot] : 0, nu = act ? T.nU[slot] : 0;
    const long ent0 = (long)T.sliceEnt[s] + i;
    int first, step, nd, nOld;
    if (B == SW_TRI_FWD || B == SW_RD) { first = 0; step = 1; nd = nl; nOld = 0; }
    else if (B == SW_TRI_BWD) { first = nl + nu - 1; step = -1; nd = nu; nOld = 0; }
    else if (B == SW_GS_FWD) { first = 0; step = 1; nd = nl; nOld = nu; }
    else { first = nl; step = 1; nd = nu; nOld = 0; }   // SW_GS_BWD
    const int M = nd + nOld;
    // what the leader needs, before any wait
    const bool leader = act && t == 0;
    double acc = 0.0, d = 1.0, rd = 0.0, sc = 0.0;
    if (act)
    {
        if (B == SW_TRI_FWD) { sc = scale[r]; acc = sc * rhs[r]; }
        else if (B == SW_TRI_BWD) { acc = w[r]; if (TF) sc = scale[r]; }
        else if (B == SW_RD) acc = scale[r];
        else { acc = rhs[r]; d = scale[r]; if (!(B == SW_GS_FWD && TF)) rd = ldu_div_prepare(d); }
    }
    int c[8];
    double v[8], v2[8], x[8];
    bool poll[8], use[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
        const int m = 8 * t + q;
        const bool dep = act && m < nd, old = act && m >= nd && m < M;
        const int k = dep ? first + m * step : nl + (m - nd);
        use[q] = dep || old;
        const long e = ent0 + (long)(use[q] ? k : 0) * LDU_WAVE;
        c[q] = use[q] ? T.col[e] : r;
        v[q] = use[q] ? val[e] : 0.0;
        v2[q] = (B == SW_RD && use[q]) ? val2[e] : 0.0;
        poll[q] = dep || (old && oldFromGranules);
        x[q] = (B == SW_RD) ? 1.0 : 0.0;
        if (old && !oldFromGranules) x[q] = w[c[q] & 0x7fffffff];
    }
    bool anyPoll = false;
#pragma unroll
    for (int q = 0; q < 8; q++) anyPoll |= poll[q];
    if (trc && lane == 0) trc[1] = (unsigned long long)wall_clock64();
    if (anyPoll)
    {
        const uint4* gp[8];
#pragma unroll
        for (int q = 0; q < 8; q++) gp[q] = SLAB ? (c[q] < 0 ? X : G) + (c[q] & 0x7fffffff) : G + c[q];
        u32x4 g[8];
        unsigned spins = 0;
        unsigned long long tw0 = 0;
        const int sleepN = g_p2p_sleep;
        const unsigned spinLimit = g_p2p_spin_limit;
        for (;;)
        {
            granule_load8(gp, g);
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 8; q++)
            {
                const unsigned want = (8 * t + q < nd) ? tag : tag - 1u;
                if (poll[q]) ok &= (g[q].y == want) & (g[q].w == want);
            }
            if (ok) break;
            if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0))
            {
                if (spins > spinLimit) p2p_dbg_record(3, r, tag, c[0], g[0].y, g[0].w, t);
                *abortFlag = 1;
                return false;
            }
            p2p_nap(spins, sleepN);
        }
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (poll[q]) x[q] = granule_value(g[q]);
    }
    // the terms, in the family's association (p2p_accumulate OP 0..3)
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
        double term;
        if (B == SW_RD) term = TF ? (v2[q] * v[q]) * (1.0 / x[q]) : (v2[q] * v[q]) / x[q];
        else if (TF && (B == SW_TRI_FWD || B == SW_TRI_BWD)) term = sc * (v[q] * x[q]);
        else term = v[q] * x[q];
        lds[(8 * t + q) * R + i] = term;
    }
    LDU_STEP_FENCE();   // LDS is in order within a wave: the leader's reads below see the helpers' writes
    if (trc && lane == 0) trc[2] = (unsigned long long)wall_clock64();
    if (leader)
    {
        double auxv = acc;
        for (int m0 = 0; m0 < 8 * Tl; m0 += 8)
        {
            double pr[8];
#pragma unroll
            for (int q = 0; q < 8; q++) pr[q] = lds[(m0 + q) * R + i];
#pragma unroll
            for (int q = 0; q < 8; q++)
            {
                if (m0 + q < M) acc -= pr[q];
                if (m0 + q < nd) auxv = acc;
            }
        }
        double out = acc;
        if (B == SW_GS_FWD)
        {
            if (aux) aux[r] = auxv;
            out = TF ? d * acc : ldu_div(acc, d, rd);
        }
        else if (B == SW_GS_BWD) out = ldu_div(acc, d, rd);
        if (writeW) w[r] = out;
        if (SLAB) granule_store_slab(G, X, r, out, tag, xflag[r] != 0);
        else granule_store(G, r, out, tag);
    }
    if (trc && lane == 0)
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trc[3] = (unsigned long long)wall_clock64(); trc[4] = xcc & 0xf; trc[5] = blockIdx.x;
    }
    return true;
}

template <bool SLAB>
__device__ __forceinline__ bool gs_upper_block(const SliceTab& T, const uint4* __restrict__ G,
                                               const uint4* __restrict__ X, bool first, unsigned t,
                                               const double* __restrict__ psi, const double* __restrict__ val,
                                               long ent, int nl, int nu, int base, bool active, int selfRow,
                                               volatile int* abortFlag, double (&pu)[8]);

template <int MODE, bool DIAG = false, bool SLAB = false>
__device__ __forceinline__ bool p2p_slice(const SliceTab& T, int s, int lane, uint4* __restrict__ G,
                                          uint4* __restrict__ X, const unsigned char* __restrict__ xflag,
                                          unsigned tag, volatile int* abortFlag, double* __restrict__ w,
                                          const double* __restrict__ rhs,
                                          const double* __restrict__ scale,
                                          const double* __restrict__ val,
                                          const double* __restrict__ val2, double* __restrict__ aux,
                                          P2PStat& waitEst, double* __restrict__ puLds = nullptr, int puSlots = 0)
{
    if (T.sliceT)
    {
        const int Tl = T.sliceT[s];   // wave-uniform
        if (Tl > 1)
        {
            ldu_debug_stall(s == 0);
            return coop_rows<MODE, SLAB, false>(T, s, Tl, lane, G, X, xflag, tag, false, true, abortFlag, w, rhs, scale, val,
                                                val2, aux, puLds);
        }
    }
    const int cnt = T.sliceCnt[s];
    if (lane >= cnt) return true;
    ldu_debug_stall(s == 0);
    const int r = T.sliceRow[s] + lane;
    const int nl = T.nL[r];
    const int nu = T.nU[r];
    const long ent = (long)T.sliceEnt[s] + lane;
    const bool exported = SLAB ? xflag[r] != 0 : false;
    double out;
    constexpr int B = sw_base(MODE);
    constexpr bool TF = sw_tform(MODE);
    if (B == SW_TRI_FWD)
    {
        const double sc = scale[r];
        double acc = sc * rhs[r];
        if (!p2p_accumulate<TF ? 2 : 0, DIAG, SLAB>(acc, G, X, tag, T.col, val, val2, ent, 0, 1, nl, r, abortFlag, waitEst, sc)) return false;
        out = acc;
    }
    else if (B == SW_TRI_BWD)
    {
        double acc = w[r];
        const double sc = TF ? scale[r] : 0.0;
        if (!p2p_accumulate<TF ? 2 : 0, DIAG, SLAB>(acc, G, X, tag, T.col, val, val2, ent, nl + nu - 1, -1, nu, r, abortFlag, waitEst, sc)) return false;
        out = acc;
    }
    else if (B == SW_RD)
    {
        double acc = scale[r];
        if (!p2p_accumulate<TF ? 3 : 1, DIAG, SLAB>(acc, G, X, tag, T.col, val, val2, ent, 0, 1, nl, r, abortFlag, waitEst)) return false;
        out = acc;
    }
    else if (B == SW_GS_FWD)
    {
        // old values of the upper neighbours: plain loads, issued before the wait
        double acc = rhs[r];
        const double d = scale[r];
        const double rd = TF ? 0.0 : ldu_div_prepare(d);
        double xu[8];
        double vu[8];
        const int nuFast = nu <= 8 ? nu : 0;
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
            if (j < nuFast)
            {
                const long e = ent + (long)(nl + j) * LDU_WAVE;
                vu[j] = val[e];
                xu[j] = w[T.col[e] & 0x7fffffff];
            }
        }
        // wider upper parts: products formed before the wait, parked in this lane's LDS slots (see p2p_gs_task)
        const bool puWide = nu > 8 && nu <= puSlots;
        if (__any(puWide))
        {
            for (int b = 0; b < puSlots && __any(puWide && nu > b); b += 8)
            {
                double pb[8];
                if (!gs_upper_block<SLAB>(T, G, X, true, 0u, w, val, ent, nl, nu, b, puWide, r, abortFlag, pb)) return false;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (puWide && b + q < nu) puLds[(b + q) * LDU_WAVE + lane] = pb[q];
            }
        }
        if (!p2p_accumulate<0, DIAG, SLAB>(acc, G, X, tag, T.col, val, val2, ent, 0, 1, nl, r, abortFlag, waitEst)) return false;
        if (aux) aux[r] = acc;
        if (nuFast)
        {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j < nuFast) acc -= vu[j] * xu[j];
        }
        else if (puWide)
        {
            for (int q = 0; q < nu; q++) acc -= puLds[q * LDU_WAVE + lane];
        }
        else
        {
            for (int k = nl; k < nl + nu; k++)
            {
                const long e = ent + (long)k * LDU_WAVE;
                acc -= val[e] * w[T.col[e] & 0x7fffffff];
            }
        }
        out = TF ? d * acc : ldu_div(acc, d, rd);
    }
    else   // SW_GS_BWD
    {
        double acc = rhs[r];
        const double d = scale[r];
        const double rd = ldu_div_prepare(d);
        if (!p2p_accumulate<0, DIAG, SLAB>(acc, G, X, tag, T.col, val, val2, ent, nl, 1, nu, r, abortFlag, waitEst)) return false;
        out = ldu_div(acc, d, rd);
    }
    w[r] = out;
    if (SLAB) granule_store_slab(G, X, r, out, tag, exported);
    else granule_store(G, r, out, tag);
    return true;
}

// XCD slabs: what a workgroup needs to find its slab's queue
struct SlabCtl {
    int nSlabs;
    int start[9];                    // slab s owns list[start[s] .. start[s+1])
    const int* list;                 // slices (or GaussSeidel tasks) of every slab in schedule order
    unsigned* tick;                  // [8] this launch's chunk tickets, one per slab (zero at launch)
    unsigned* tickNext;              // [8] the next launch's: zeroed by this one
    unsigned* done;                  // [8] chunks reported complete, per slab (run-ahead window), zero at launch
    unsigned* doneNext;              // [8] the next launch's
    int window[8];                   // run-ahead window per slab in chunks (0 = none)
    uint4* X;                        // write-through granule copies of exported rows
    const unsigned char* xflag;      // per row: exported
};

__device__ __forceinline__ int xcc_id()
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return (int)(xcc & 0xf);
}

// One sweep on the XCD-slab engine: the workgroups that landed on XCD x serve slab x's queue (slices
// in level order; reversed for backward sweeps) - everything else is sweep_p2p_kernel.  Progress:
// the queues are restrictions of ONE global topological order and every workgroup of the grid is
// resident, so the globally first unfinished slice is always held by a running wave.
template <int MODE, bool DESC>
__global__ void __launch_bounds__(P2P_BLK)
sweep_slab_kernel(SliceTab T, SlabCtl C, uint4* G, unsigned tag, int* abortFlag, double* w,
                  const double* rhs, const double* scale, const double* val, const double* val2, double* aux,
                  int puSlots)
{
    __shared__ int s_chunk[2];
    extern __shared__ double s_pu[];   // [wave][puSlots][64] (GaussSeidel rows with wide upper parts)
    const int slab = xcc_id();
    if (slab >= C.nSlabs) return;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int first = C.start[slab];
    const int nSl = C.start[slab + 1] - first;
    const int nChunks = (nSl + P2P_CHUNK - 1) / P2P_CHUNK;
    unsigned* ticket = C.tick + slab;
    P2PStat waitEst = {0, 0, 0, -1, nullptr};
    int nextT = 0;
    const int window = C.window[slab];
    unsigned* const done = C.done + slab;
    if (threadIdx.x == 0)
    {
        __hip_atomic_store(C.tickNext + slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(C.doneNext + slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nextT = (int)atomicAdd(ticket, 1u);
    }
    for (int it = 0;; it++)
    {
        if (threadIdx.x == 0)
        {
            if (window && it) atomicAdd(done, 1u);   // the previous chunk is complete (barrier at the end of the loop body)
            const int t = ldu_abort_seen(abortFlag, it) ? 0x7fffffff : nextT;
            if (window && t < nChunks) p2p_window_wait(done, 0u, t, window, abortFlag);
            s_chunk[it & 1] = t;
            if (t < nChunks) nextT = (int)atomicAdd(ticket, 1u);
        }
        __syncthreads();
        const int chunk = s_chunk[it & 1];
        if (chunk >= nChunks) return;
        const int si = chunk * P2P_CHUNK + wave;
        if (si < nSl)
        {
            const int s = C.list[first + (DESC ? nSl - 1 - si : si)];
            p2p_slice<MODE, false, true>(T, s, lane, G, C.X, C.xflag, tag, abortFlag, w, rhs, scale, val, val2,
                                         aux, waitEst, s_pu + (size_t)wave * puSlots * LDU_WAVE, puSlots);
        }
        if (window) __syncthreads();
    }
}

template <int MODE, bool DESC, bool DIAG>
__global__ void __launch_bounds__(P2P_BLK)
sweep_p2p_kernel(SliceTab T, int nSlices, int nChunks, unsigned* ticket, unsigned ticketBase, int window,
                 unsigned doneBase, uint4* G,
                 unsigned tag, int* abortFlag, double* w, const double* rhs, const double* scale,
                 const double* val, const double* val2, double* aux, int puSlots)
{
    unsigned* const done = ticket + 32;   // chunks reported complete (own cache line)
    __shared__ int s_chunk[2];
    extern __shared__ double s_pu[];   // [wave][puSlots][64] (GaussSeidel rows with wide upper parts)
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    P2PStat waitEst = {0, 0, 0, -1, nullptr};   // gate + timestamps / poll count of the current slice
    // thread 0 keeps one ticket in flight ahead of the one being processed
    int nextT = 0;
    if (threadIdx.x == 0) nextT = (int)(atomicAdd(ticket, 1u) - ticketBase);
    for (int it = 0;; it++)
    {
        if (threadIdx.x == 0)
        {
            if (window && it) atomicAdd(done, 1u);
            // an expired spin anywhere drains the whole grid (uniform per workgroup)
            const int t = ldu_abort_seen(abortFlag, it) ? 0x7fffffff : nextT;
            if (window && t < nChunks) p2p_window_wait(done, doneBase, t, window, abortFlag);
            s_chunk[it & 1] = t;
            if (t < nChunks) nextT = (int)(atomicAdd(ticket, 1u) - ticketBase);
        }
        __syncthreads();
        const int chunk = s_chunk[it & 1];
        if (chunk >= nChunks) return;
        const int si = chunk * P2P_CHUNK + wave;
        if (si < nSlices)
        {
            const int s = DESC ? nSlices - 1 - si : si;
            const long long tT = (DIAG && g_p2p_trace) ? wall_clock64() : 0;
            if (DIAG)
            {
                waitEst.polls = 0;
                waitEst.gateSlice = T.gate ? T.gate[s] : -1;
                waitEst.sliceDone = T.sliceDone;
            }
            p2p_slice<MODE, DIAG, false>(T, s, lane, G, nullptr, nullptr, tag, abortFlag, w, rhs, scale, val, val2,
                                         aux, waitEst, s_pu + (size_t)wave * puSlots * LDU_WAVE, puSlots);
            if (DIAG && T.sliceDone && lane == 0)
                __hip_atomic_store(T.sliceDone + s, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (DIAG && g_p2p_trace && lane == 0)
            {
                unsigned long long* rec = g_p2p_trace + (size_t)s * 8;
                unsigned xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                rec[0] = (unsigned long long)tT; rec[1] = (unsigned long long)waitEst.tWait;
                rec[2] = (unsigned long long)waitEst.tReady; rec[3] = (unsigned long long)wall_clock64();
                rec[4] = waitEst.polls; rec[5] = xcc & 0xf; rec[6] = (unsigned long long)blockIdx.x; rec[7] = chunk;
            }
        }
        if (window) __syncthreads();
    }
}

// ---- self-check of ldu_div (ldu_internal.hpp) against the compiler's division: ldu_debug_div_check
__device__ __forceinline__ unsigned long long dc_mix(unsigned long long z)
{
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double dc_make(unsigned long long bits, int mode, int which)
{
    // mode (uniform per wave, so that the wave-level choice in ldu_div is exercised both ways):
    // 0 any bit pattern; 1 exponents around the edges of the fast range; 2 moderate exponents (always fast);
    // 3 moderate denominator, numerator zero / -0 / denormal / tiny / huge
    unsigned long long m = bits & 0x800fffffffffffffull;
    unsigned long long e;
    if (mode == 0) return __longlong_as_double((long long)bits);
    if (mode == 1) { const int edge[4] = {697, 700, 1300, 1303}; e = (unsigned long long)(edge[(bits >> 52) & 3] + (int)((bits >> 54) & 3) - 1); }
    else if (mode == 2 || which == 1) e = 1023ull - 60ull + ((bits >> 52) % 121ull);
    else
    {
        const int sel = (int)((bits >> 52) & 7);
        if (sel == 0) return 0.0;
        if (sel == 1) return -0.0;
        if (sel == 2) { e = 0; }                       // denormal
        else if (sel == 3) e = 1ull + ((bits >> 55) & 63);
        else if (sel == 4) e = 2046ull - ((bits >> 55) & 63);
        else e = 1023ull - 400ull + ((bits >> 55) % 801ull);
    }
    return __longlong_as_double((long long)(m | (e << 52)));
}
__global__ void __launch_bounds__(256) div_check_kernel(unsigned long long seed, long n, unsigned long long* out)
{
    unsigned long long bad = 0;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    {
        const int mode = (int)((i >> 6) & 3);
        const double t = dc_make(dc_mix(seed + 2ull * (unsigned long long)i), mode, 0);
        const double d = dc_make(dc_mix(seed + 2ull * (unsigned long long)i + 1ull), mode, 1);
        const double r = ldu_div_prepare(d);
        const double q = ldu_div(t, d, r);
        double tt = t, ddv = d;
        asm volatile("" : "+v"(tt), "+v"(ddv));       // an independent division, not a copy of the one inside ldu_div
        const double ref = tt / ddv;
        const bool same = __double_as_longlong(q) == __double_as_longlong(ref) || (q != q && ref != ref);
        if (!same) bad++;
    }
    if (bad) atomicAdd(out, bad);
}
int k_div_check(ldu_ctx* ctx, unsigned long long seed, long n, unsigned long long* mismatches)
{
    unsigned long long* d = nullptr;
    LDU_CHECK_HIP(hipMalloc((void**)&d, sizeof(unsigned long long)));
    LDU_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
    div_check_kernel<<<ctx->numCUs * 8, 256, 0, ctx->stream>>>(seed, n, d);
    LDU_CHECK_HIP(hipMemcpyAsync(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    return 0;
}

// ---- achievable HBM bandwidth of this chip, measured with the library's own stream kernels (ldu_debug_stream):
// copy a[i] = b[i] (16 B per element) and triad a[i] = b[i] + s*c[i] (24 B per element, McCalpin's STREAM), f64,
// grid-stride over numCUs x 8 workgroups of 256 threads, two elements per thread and trip (dwordx4 accesses).
// bench.py prints the triad figure as roofline.peak_measured next to the 8 TB/s spec (SURVEY.md 8d "bound").
typedef double stream_d2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) stream_kernel(long n2, stream_d2* __restrict__ a, const stream_d2* __restrict__ b,
                                                     const stream_d2* __restrict__ c, double s)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
    {
        const stream_d2 x = __builtin_nontemporal_load(b + i);
        stream_d2 y;
        if (MODE == 0) y = x;
        else
        {
            const stream_d2 z = __builtin_nontemporal_load(c + i);
            y.x = x.x + s * z.x; y.y = x.y + s * z.y;
        }
        __builtin_nontemporal_store(y, a + i);
    }
}

// mode 0 copy, 1 triad; n doubles per array (rounded down to even); returns the average seconds of `reps` launches
// (HIP events on the library's stream, after one untimed launch)
int k_stream(ldu_ctx* ctx, int mode, long n, int reps, double* seconds)
{
    const long n2 = n / 2;
    if (n2 < 1 || reps < 1 || mode < 0 || mode > 1) { ldu_set_error("ldu_debug_stream: bad argument"); return -1; }
    stream_d2 *a = nullptr, *b = nullptr, *c = nullptr;
    LDU_CHECK_HIP(hipMalloc((void**)&a, sizeof(stream_d2) * n2));
    LDU_CHECK_HIP(hipMalloc((void**)&b, sizeof(stream_d2) * n2));
    LDU_CHECK_HIP(hipMalloc((void**)&c, sizeof(stream_d2) * n2));
    LDU_CHECK_HIP(hipMemsetAsync(b, 0, sizeof(stream_d2) * n2, ctx->stream));
    LDU_CHECK_HIP(hipMemsetAsync(c, 0, sizeof(stream_d2) * n2, ctx->stream));
    hipEvent_t e0, e1;
    LDU_CHECK_HIP(hipEventCreate(&e0));
    LDU_CHECK_HIP(hipEventCreate(&e1));
    const int grid = (int)std::min<long>((n2 + 255) / 256, (long)ctx->numCUs * 8);
    for (int r = -1; r < reps; r++)
    {
        if (r == 0) LDU_CHECK_HIP(hipEventRecord(e0, ctx->stream));
        if (mode == 0) stream_kernel<0><<<grid, 256, 0, ctx->stream>>>(n2, a, b, c, 3.0);
        else stream_kernel<1><<<grid, 256, 0, ctx->stream>>>(n2, a, b, c, 3.0);
    }
    LDU_CHECK_HIP(hipEventRecord(e1, ctx->stream));
    LDU_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    LDU_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *seconds = (double)ms * 1e-3 / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
    return 0;
}

// Placement census: the slab engine needs workgroups of one launch on every XCD it assigns a slab to
// and HW_REG_XCC_ID values 0..n-1.  HIP promises neither, so it is checked once per context with the
// sweep kernels' own geometry; a failed census leaves the chip-wide engine in charge.
__global__ void __launch_bounds__(P2P_BLK) xcc_census_kernel(int* out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

int k_xcd_census(ldu_ctx* ctx)
{
    const int grid = ctx->numCUs * ctx->p2pBlocksPerCU;
    int* d = nullptr;
    LDU_CHECK_HIP(hipMalloc((void**)&d, sizeof(int) * (size_t)grid));
    xcc_census_kernel<<<grid, P2P_BLK, 0, ctx->stream>>>(d);
    std::vector<int> h(grid);
    LDU_CHECK_HIP(hipMemcpyAsync(h.data(), d, sizeof(int) * (size_t)grid, hipMemcpyDeviceToHost, ctx->stream));
    LDU_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    int cnt[16] = {0};
    for (int v : h) cnt[v & 15]++;
    int n = 0;
    while (n < 16 && cnt[n] > 0) n++;
    bool ok = n >= 1 && n <= 8;
    for (int i = n; i < 16; i++) ok = ok && cnt[i] == 0;
    for (int i = 0; i < n; i++) ok = ok && cnt[i] * n * 2 >= grid;   // at least half its fair share
    ctx->nXcd = ok ? n : 0;
    if (getenv("LDU_VERBOSE"))
        fprintf(stderr, "[ldugpu] XCD census: %d workgroups on %d XCDs -> slab engine %s\n", grid, n,
                ok ? "available" : "disabled");
    return 0;
}

// the slabs' level ranges overlap little: sum of the spans < 3 x the number of levels (a natural numbering, where the
// slabs cut across the level planes: ~ nSlabs x; bandCompression-numbered graph and its agglomerated levels: 1.3 - 2.4 x)
static bool slabs_sequential(const ldu_addr* a)
{
    if (a->nSlabs < 2) return false;
    long sum = 0;
    for (int i = 0; i < a->nSlabs; i++) sum += a->slabLevelSpan[i];
    static const double factor = getenv("LDU_SLAB_SEQ_FACTOR") ? atof(getenv("LDU_SLAB_SEQ_FACTOR")) : 3.0;
    return (double)sum < factor * a->nLevels;
}

// Workgroups per CU of a slab-engine launch with k sweeps in flight.  Every waiting wave slows the
// hand-offs of its CU (measured: 0.86 us per level at 1 workgroup per CU, 1.0 at 2, 1.1 at 3), but
// too few waves cannot cover the several dependent loads a slice needs before it can wait: about
// 8 levels of look-ahead are needed (tools/det_probe.py).
static int slab_bpc(const ldu_addr* a, int k)
{
    const ldu_ctx* ctx = a->ctx;
    {
        static const char* e = getenv("LDU_SLAB_BPC");   // experiment knob: workgroups per CU of every slab launch
        if (e && atoi(e) > 0) return std::min(atoi(e), ctx->p2pMaxBlocksPerCU);
    }
    if (ctx->p2pBpcForced) return std::min(ctx->p2pBlocksPerCU * k, ctx->p2pMaxBlocksPerCU);
    const double wavesPerXcdPerBpc = std::max(1, ctx->numCUs / std::max(1, ctx->nXcd)) * 4.0;
    if (slabs_sequential(a))
    {
        // one slab (one XCD) carries a whole dependency level, and the k sweeps sit in different slabs: two levels
        // of look-ahead per XCD; more waiting waves only slow the hand-offs (irregular 216^3 graph, 44 slices per
        // level: 4 sweeps 15.9 / 22.0 / 32.5 ms at 1 / 2 / 3 workgroups per CU)
        const double levelWidth = (double)a->nCells / LDU_WAVE / std::max(1, a->nLevels);
        const int b = (int)std::ceil(2.0 * levelWidth / wavesPerXcdPerBpc);
        return std::max(1, std::min(b, std::min(4, ctx->p2pMaxBlocksPerCU)));
    }
    int bpc = (int)std::ceil(8.0 * k * a->slabWidth / wavesPerXcdPerBpc);
    return std::max(1, std::min(bpc, std::min(4, ctx->p2pMaxBlocksPerCU)));
}

// Which engine runs a sweep.  The slab engine's hand-off is faster (same-XCD L2), but on wide
// levels both engines are bound by the CUs' memory queues (the streaming loads of the rows ahead
// delay the polls of the rows at the front) and the slab engine's fixed slab->XCD binding balances
// worse: measured break-even widths (average slices per level per slab), tools/det_probe.py.
// kind: 0 = triangular sweeps, 1 = one GaussSeidel sweep, 2 = k pipelined GaussSeidel sweeps
// (k sweeps in flight multiply the width; with k > 2 the chip-wide engine's 8x larger pool of
//  waves wins except on one-XCD-sized matrices: per-level table of the 216^3 GAMG hierarchy in
//  profiles/r01_xcd_slab_probe.md)
static bool use_slab(const ldu_addr* a, int kind, int k = 1)
{
    const ldu_ctx* ctx = a->ctx;
    if (a->nSlabs <= 0 || ctx->p2pTrace) return false;
    if (ctx->p2pSlabs > 0) return true;   // forced
    if (kind == 0) return a->slabWidth <= 24.0;
    if (kind == 1) return a->slabWidth <= 16.0;
    if (k <= 2) return k * a->slabWidth <= 14.0;
    // Slabs that follow one another along the dependency levels (numberings whose index ranges follow the levels:
    // bandCompression): every level lies in ONE slab, the k sweeps of a launch sit in k different slabs most of the
    // time, each front on its own XCD with same-XCD hand-offs.  The chip-wide engine, with all k fronts polled for
    // by the whole chip, took 9-11 us per level there (irregular 216^3 graph, tools/gsm_trace.py: 4 sweeps 40.8 ms).
    if (slabs_sequential(a)) return (double)a->nCells / LDU_WAVE / std::max(1, a->nLevels) <= 96.0;
    if (a->lagBuckets && a->nCells <= 150000) return true;   // (see choose_slabs: eight slabs on mid-size irregular levels)
    return a->nSlabs == 1 && k * a->slabWidth <= 10.0;
}

// which engine serves a sweep kind on this addressing (introspection for bench.py / tests):
// 0 chip-wide point-to-point, 1 XCD slabs, 2 clusters, 3 single wavefront (tiny), 4 level kernels, 5 one workgroup (LDS),
// 6 blocks (LDS-resident blocks of a few thousand cells, ldu_blocks.hip; pipelined GaussSeidel sweeps only)
int k_engine_of(ldu_addr* a, int kind /* 0 triangular, 1 one GaussSeidel sweep, 2 two pipelined sweeps */)
{
    ldu_ctx* ctx = a->ctx;
    if (!ctx->sweepP2P) return 4;
    // (with coupled patches the cluster engine sweeps one by one; the block engine pipelines cyclic interfaces: ldu_solvers.cpp smooth_gs)
    if (kind == 2 && (a->nPatchFaces || !(ctx->clusterMulti && k_cluster_active(a))) && k_blocks_active(a)) return 6;
    if (kind >= 1 && !a->nPatchFaces && ctx->wgEngine && a->wgLevel && 9 * (size_t)a->nCells + 64 <= 160 * 1024) return 5;
    if (kind >= 1 && ctx->smallKernels && a->maxRowWidth <= 16 && !a->nPatchFaces
        && a->nCells <= ((kind == 2 && ctx->smallPipe) ? ctx->smallMaxCells : std::min(ctx->smallMaxCells, 3000)))
        return 3;
    if (kind == 2 ? (ctx->clusterMulti && k_cluster_active(a)) : k_cluster_kind_active(a, kind)) return 2;
    return use_slab(a, kind, 2) ? 1 : 0;
}

// per-launch part of SlabCtl: flips the ticket parity (this launch's counters were zeroed by the
// previous launch on this lane, or by the allocation)
static void slab_ctl(ldu_addr* a, ldu_addr::P2PLane& P, SlabCtl& C)
{
    C.nSlabs = a->nSlabs;
    C.tick = P.d_ctl + 8 * P.par;
    C.tickNext = P.d_ctl + 8 * (P.par ^ 1u);
    C.done = P.d_ctl + 16 + 8 * P.par;
    C.doneNext = P.d_ctl + 16 + 8 * (P.par ^ 1u);
    for (int i = 0; i < 8; i++) C.window[i] = 0;
    P.par ^= 1u;
    C.X = P.d_X;
    C.xflag = a->d_xflag;
}

// run-ahead window in chunks: `levels` dependency levels' worth of chunks of k sweeps; 0 (off) when the grid could not
// run further ahead than that anyway
static int p2p_window(const ldu_addr* a, long nChunks, int k, int grid, int nLevelsOfQueue = 0)
{
    const double lv = a->ctx->p2pWindowLevels;
    if (nLevelsOfQueue <= 0) nLevelsOfQueue = a->nLevels;
    if (lv <= 0 || nLevelsOfQueue <= 0) return 0;
    // (a slab's queue only spans the slab's levels: measured against ALL levels of the addressing the window of a
    //  bandCompression-numbered mesh was one level instead of eight - irregular 216^3: one sweep 21.7 instead of 13.8 ms)
    const double perLevel = (double)nChunks / (double)nLevelsOfQueue;    // chunks per level, all k sweeps together
    long w = (long)(lv * perLevel + 0.5);
    if (w < 8 * k) w = 8 * k;
    // The window only pays when the grid would otherwise run dozens of levels ahead of the front (deep, narrow DAGs:
    // every resident wave polls for data that is far away).  On levels a quarter of the grid wide or wider the
    // bookkeeping (one atomic and a workgroup barrier per chunk, workgroups idling at the window's edge) costs more
    // than the polls it saves: octree twin of the motorBike mesh, finest level (660 levels, 59 chunks each, grid 512):
    // one GaussSeidel sweep 4.9 ms with the window, 1.9 ms without; two pipelined 9.3 / 3.5 ms; DIC 9.8 / 3.1 ms.
    // (round 2's rule was w >= 2 grid; measured per GAMG level of that mesh: off wins at >= 2.5 chunks per level and
    //  slab (64 workgroups per slab), on wins below 1.2)
    if ((double)grid <= 2.0 * lv * perLevel) return 0;   // the grid cannot run more than 2 windows (16 levels) ahead
    return (int)w;
}

// LDS slots per lane for the parked upper-part products of rows with more than eight upper neighbours (0: none)
static int gs_pu_slots(const ldu_addr* a)
{
    const int coop = a->nCoopSlices ? 8 : 0;   // cooperative rows: 512 doubles per wave
    if (a->maxUpper <= 8 || !a->ctx->gsWideUpper) return coop;
    return std::max(coop, std::min(24, (a->maxUpper + 7) / 8 * 8));
}

template <int MODE, bool DESC>
static int launch_p2p(ldu_addr* a, const SweepArgs& g, hipStream_t s)
{
    ldu_ctx* ctx = a->ctx;
    ldu_addr::P2PLane* Pp = a->lane(g.lane);
    if (!Pp) { ldu_set_error("p2p lane allocation failed"); return -1; }
    ldu_addr::P2PLane& P = *Pp;
    // forward GaussSeidel sweeps park the products of wide upper parts in LDS (see p2p_gs_task)
    const int puSlots = sw_base(MODE) == SW_GS_FWD ? gs_pu_slots(a) : (a->nCoopSlices ? 8 : 0);
    const size_t puBytes = sizeof(double) * (size_t)(P2P_BLK / LDU_WAVE) * puSlots * LDU_WAVE;
    if (use_slab(a, (sw_base(MODE) == SW_GS_FWD || sw_base(MODE) == SW_GS_BWD) ? 1 : 0))
    {
        SliceTab TS{a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_nL, a->d_nU, a->d_colX};
        if (a->nCoopSlices) TS.sliceT = a->d_sliceT;
        SlabCtl C;
        slab_ctl(a, P, C);
        C.list = a->d_slabList;
        for (int i = 0; i <= 8; i++) C.start[i] = a->slabStart[i];
        P.epoch++;
        if (P.epoch == 0) P.epoch = 1;
        const int grid = ctx->numCUs * (puSlots > 8 ? std::min(slab_bpc(a, 1), 3) : slab_bpc(a, 1));
        for (int i = 0; i < a->nSlabs; i++)
            C.window[i] = p2p_window(a, cdiv(a->slabStart[i + 1] - a->slabStart[i], P2P_CHUNK), 1,
                                     grid / std::max(1, a->nSlabs), a->slabLevelSpan[i]);
        sweep_slab_kernel<MODE, DESC><<<grid, P2P_BLK, puBytes, s>>>(TS, C, P.d_granule, P.epoch, ctx->d_abort, g.w,
            g.rhs, g.scale, g.val, g.val2, g.aux, puSlots);
        LDU_CHECK_HIP(hipGetLastError());
        return 0;
    }
    SliceTab T{a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_nL, a->d_nU, a->d_col};
    if (a->nCoopSlices) T.sliceT = a->d_sliceT;
    if (ctx->p2pGate)
    {
        T.gate = DESC ? a->d_gateB : a->d_gateF;
        T.sliceDone = a->d_sliceDone;
    }
    const int nChunks = cdiv(a->nSlices, P2P_CHUNK);
    int grid = ctx->numCUs * ctx->p2pBlocksPerCU * 256 / P2P_BLK;   // p2pBlocksPerCU counts 256-thread units
    if (grid > nChunks) grid = nChunks;
    if (grid < 1) grid = 1;
    if (P.gen != ctx->p2pGen)
    {
        // a previous sweep aborted somewhere: ticket counters are no longer in step
        LDU_CHECK_HIP(hipMemsetAsync(P.d_ticket, 0, sizeof(unsigned) * 64, s));
        P.ticketBase = 0;
        P.doneBase = 0;
        P.gen = ctx->p2pGen;
    }
    P.epoch++;
    if (P.epoch == 0) P.epoch = 1;   // tag 0 = never published
    const int window = p2p_window(a, nChunks, 1, grid);
    if (ctx->p2pGate || ctx->p2pTrace)
        sweep_p2p_kernel<MODE, DESC, true><<<grid, P2P_BLK, puBytes, s>>>(T, a->nSlices, nChunks, P.d_ticket,
            P.ticketBase, window, P.doneBase, P.d_granule, P.epoch, ctx->d_abort, g.w, g.rhs, g.scale, g.val, g.val2, g.aux, puSlots);
    else
        sweep_p2p_kernel<MODE, DESC, false><<<grid, P2P_BLK, puBytes, s>>>(T, a->nSlices, nChunks, P.d_ticket,
            P.ticketBase, window, P.doneBase, P.d_granule, P.epoch, ctx->d_abort, g.w, g.rhs, g.scale, g.val, g.val2, g.aux, puSlots);
    // every workgroup overshoots the ticket exactly once; with a window every chunk is reported complete once
    P.ticketBase += (unsigned)(nChunks + grid);
    if (window) P.doneBase += (unsigned)nChunks;
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// Debug: per-task timeline of the pipelined GaussSeidel sweeps of the level engines (ldu_debug_gs_multi_trace):
// 8 x u64 per (sweep, slice): tStart, tUpperDone, tLowerDone, tStored [100 MHz wall clock], XCC id, workgroup, 0, 0
__device__ unsigned long long* g_gsm_trace = nullptr;
__device__ int g_gsm_trace_stride = 0;
static bool g_wg_trace_on = false;   // host copy: the one-workgroup engine launches its traced variant
int k_set_gs_multi_trace(unsigned long long* buf, int nSlices)
{
    g_wg_trace_on = buf != nullptr;
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gsm_trace), &buf, sizeof(buf)));
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gsm_trace_stride), &nSlices, sizeof(int)));
    return 0;
}

// ---------------------------------------------------------------- pipelined Gauss-Seidel sweeps
// four "old" (previous sweep) neighbour values xu[BASE..BASE+3] of a row, static indices only
template <int BASE, bool SLAB = false>
__device__ __forceinline__ bool gs_gather_old4(const SliceTab& T, const uint4* __restrict__ G,
                                               const uint4* __restrict__ X, unsigned t,
                                               const double* __restrict__ val, long ent, int nl, int nu,
                                               int selfRow, volatile int* abortFlag, double (&xu)[8],
                                               double (&vu)[8])
{
    int c[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const bool need = BASE + q < nu;
        const long e = ent + (long)(nl + BASE + q) * LDU_WAVE;
        c[q] = need ? T.col[e] : selfRow;
        vu[BASE + q] = need ? val[e] : 0.0;
    }
    const uint4* gp[4];
#pragma unroll
    for (int q = 0; q < 4; q++) gp[q] = SLAB ? (c[q] < 0 ? X : G) + (c[q] & 0x7fffffff) : G + c[q];
    u32x4 g0, g1, g2, g3;
    unsigned spins = 0;
            unsigned long long tw0 = 0;
    const unsigned spinLimit = g_p2p_spin_limit;
    for (;;)
    {
        granule_load4(gp[0], gp[1], gp[2], gp[3], g0, g1, g2, g3);
        bool ok = true;
        if (BASE + 0 < nu) ok &= (g0.y == t) & (g0.w == t);
        if (BASE + 1 < nu) ok &= (g1.y == t) & (g1.w == t);
        if (BASE + 2 < nu) ok &= (g2.y == t) & (g2.w == t);
        if (BASE + 3 < nu) ok &= (g3.y == t) & (g3.w == t);
        if (ok) break;
        if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0))
        {
            if (spins > spinLimit)
            {
                if (BASE + 0 < nu && !((g0.y == t) & (g0.w == t))) p2p_dbg_record(2, selfRow, t, c[0], g0.y, g0.w, BASE);
                else if (BASE + 1 < nu && !((g1.y == t) & (g1.w == t))) p2p_dbg_record(2, selfRow, t, c[1], g1.y, g1.w, BASE + 1);
                else if (BASE + 2 < nu && !((g2.y == t) & (g2.w == t))) p2p_dbg_record(2, selfRow, t, c[2], g2.y, g2.w, BASE + 2);
                else if (BASE + 3 < nu) p2p_dbg_record(2, selfRow, t, c[3], g3.y, g3.w, BASE + 3);
            }
            *abortFlag = 1;
            return false;
        }
        // never on the critical path (the previous sweep runs ahead): back off quickly so that
        // thousands of waiting waves do not starve the waves of the sweep they wait for
        p2p_nap(spins, 1);
    }
    xu[BASE + 0] = granule_value(g0);
    xu[BASE + 1] = granule_value(g1);
    xu[BASE + 2] = granule_value(g2);
    xu[BASE + 3] = granule_value(g3);
    return true;
}

// the same for rows with 5..8 upper neighbours: eight columns, then eight granules, each in ONE round trip
template <bool SLAB = false>
__device__ __forceinline__ bool gs_gather_old8(const SliceTab& T, const uint4* __restrict__ G,
                                               const uint4* __restrict__ X, unsigned t,
                                               const double* __restrict__ val, long ent, int nl, int nu,
                                               int selfRow, volatile int* abortFlag, double (&xu)[8],
                                               double (&vu)[8])
{
    int c[8];
    const uint4* gp[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
        const bool need = q < nu;
        const long e = ent + (long)(nl + q) * LDU_WAVE;
        c[q] = need ? T.col[e] : selfRow;
        vu[q] = need ? val[e] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) gp[q] = SLAB ? (c[q] < 0 ? X : G) + (c[q] & 0x7fffffff) : G + c[q];
    u32x4 g[8];
    unsigned spins = 0;
            unsigned long long tw0 = 0;
    const unsigned spinLimit = g_p2p_spin_limit;
    for (;;)
    {
        granule_load8(gp, g);
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (q < nu) ok &= (g[q].y == t) & (g[q].w == t);
        if (ok) break;
        if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0))
        {
            if (spins > spinLimit)
            {
                bool rec = false;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (!rec && q < nu && !((g[q].y == t) & (g[q].w == t)))
                    {
                        p2p_dbg_record(2, selfRow, t, c[q], g[q].y, g[q].w, q);
                        rec = true;
                    }
            }
            *abortFlag = 1;
            return false;
        }
        p2p_nap(spins, 1);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) xu[q] = granule_value(g[q]);
    return true;
}

// Rows with more than eight upper neighbours (agglomerated levels of an irregular mesh: 10-15 of them): the products
// coefficient x old value of entries base .. base+7 of the upper part, for the lanes with `active`.  Sweep 0 reads the
// old psi, a later sweep the previous sweep's granules (tag t).
template <bool SLAB>
__device__ __forceinline__ bool gs_upper_block(const SliceTab& T, const uint4* __restrict__ G,
                                               const uint4* __restrict__ X, bool first, unsigned t,
                                               const double* __restrict__ psi, const double* __restrict__ val,
                                               long ent, int nl, int nu, int base, bool active, int selfRow,
                                               volatile int* abortFlag, double (&pu)[8])
{
    int c[8];
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
        const bool need = active && base + q < nu;
        const long e = ent + (long)(nl + base + q) * LDU_WAVE;
        c[q] = need ? T.col[e] : selfRow;
        v[q] = need ? val[e] : 0.0;
    }
    if (first)
    {
        double x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = psi[c[q] & 0x7fffffff];
#pragma unroll
        for (int q = 0; q < 8; q++) pu[q] = v[q] * x[q];
        return true;
    }
    const uint4* gp[8];
#pragma unroll
    for (int q = 0; q < 8; q++) gp[q] = SLAB ? (c[q] < 0 ? X : G) + (c[q] & 0x7fffffff) : G + c[q];
    u32x4 g[8];
    unsigned spins = 0;
            unsigned long long tw0 = 0;
    const unsigned spinLimit = g_p2p_spin_limit;
    for (;;)
    {
        granule_load8(gp, g);
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (active && base + q < nu) ok &= (g[q].y == t) & (g[q].w == t);
        if (ok) break;
        if (ldu_wait_expired(spins, spinLimit, abortFlag, tw0))
        {
            if (spins > spinLimit)
            {
                bool rec = false;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (!rec && active && base + q < nu && !((g[q].y == t) & (g[q].w == t)))
                    {
                        p2p_dbg_record(2, selfRow, t, c[q], g[q].y, g[q].w, base + q);
                        rec = true;
                    }
            }
            *abortFlag = 1;
            return false;
        }
        p2p_nap(spins, 1);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) pu[q] = v[q] * granule_value(g[q]);
    return true;
}

// k consecutive GaussSeidel sweeps of the SAME matrix in ONE launch.  Sweep j+1 of a row only needs
// sweep j's values of its UPPER neighbours (its "old" values) and sweep j+1's values of its LOWER
// neighbours, so sweep j+1 can trail sweep j by the level distance to the upper neighbours: the
// whole smoothing takes (nLevels + skew*(k-1)) hand-offs instead of k*nLevels.  Arithmetic per row
// is untouched (GaussSeidelSmoother.C:151-176), so the result is bit-identical to k separate sweeps.
// Tasks (sweep, slice) are ticketed in a host-built topological order (plan_gs_tasks); tag E+j marks
// "value of sweep j".  Only the last sweep writes psi (earlier values live in the granules).
template <bool SLAB>
__device__ __forceinline__ void p2p_gs_task(const SliceTab& T, int s, int j, int k, int lane,
                                            uint4* __restrict__ G, uint4* __restrict__ X,
                                            const unsigned char* __restrict__ xflag, unsigned tag0,
                                            volatile int* abortFlag, double* __restrict__ psi,
                                            const double* __restrict__ rhs,
                                            const double* __restrict__ diag,
                                            const double* __restrict__ val, P2PStat& waitEst,
                                            double* __restrict__ puLds = nullptr, int puSlots = 0)
{
    if (T.sliceT)
    {
        const int Tl = T.sliceT[s];   // wave-uniform
        if (Tl > 1)
        {
            ldu_debug_stall(s == 0 && j == 0);
            (void)coop_rows<SW_GS_FWD, SLAB, true>(T, s, Tl, lane, G, X, xflag, tag0 + (unsigned)j, j > 0, j == k - 1, abortFlag,
                                                   psi, rhs, diag, val, nullptr, nullptr, puLds,
                                                   g_gsm_trace ? g_gsm_trace + ((size_t)j * (size_t)g_gsm_trace_stride + (size_t)s) * 8 : nullptr);
            return;
        }
    }
    const int cnt = T.sliceCnt[s];
    if (lane >= cnt) return;
    ldu_debug_stall(s == 0 && j == 0);
    unsigned long long* const trc = g_gsm_trace ? g_gsm_trace + ((size_t)j * (size_t)g_gsm_trace_stride + (size_t)s) * 8 : nullptr;
    if (trc && lane == 0) trc[0] = (unsigned long long)wall_clock64();
    const int slot = T.sliceRow[s] + lane;
    const int r = T.rowIdx ? T.rowIdx[slot] : slot;
    const int nl = T.nL[slot];
    const int nu = T.nU[slot];
    const long ent = (long)T.sliceEnt[s] + lane;
    const bool exported = SLAB ? xflag[r] != 0 : false;
    double acc = rhs[r];
    const double d = diag[r];
    const double rd = ldu_div_prepare(d);
    const unsigned tagNew = tag0 + (unsigned)j;
    // 1. "old" values of the upper neighbours: before the kernel (sweep 0) or sweep j-1's granules,
    //    which are long published (that sweep runs ahead of this one)
    double xu[8], vu[8];
    const int nuFast = nu <= 8 ? nu : -1;
    // Wider upper parts: their products are formed HERE, before the wait for the lower neighbours, and parked in this
    // lane's LDS slots (subtracted in face order after the lower part, as the reference does).  Left to the end - one
    // dependent column / value round trip per entry after the lower neighbours had arrived - they were most of a
    // dependency level's time on agglomerated levels (20-29 neighbours per row: 8-20 us per level).
    const bool puWide = nuFast < 0 && nu <= puSlots;
    if (__any(puWide))
    {
        for (int b = 0; b < puSlots && __any(puWide && nu > b); b += 8)
        {
            double pb[8];
            if (!gs_upper_block<SLAB>(T, G, X, j == 0, tagNew - 1u, psi, val, ent, nl, nu, b, puWide, r, abortFlag, pb)) return;
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (puWide && b + q < nu) puLds[(b + q) * LDU_WAVE + lane] = pb[q];
        }
    }
    if (nuFast >= 0)
    {
        if (j == 0)
        {
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (q < nuFast)
                {
                    const long e = ent + (long)(nl + q) * LDU_WAVE;
                    vu[q] = val[e];
                    xu[q] = psi[T.col[e] & 0x7fffffff];
                }
        }
        else
        {
            if (g_p2p_wide && __any(nuFast > 4))
            {
                if (!gs_gather_old8<SLAB>(T, G, X, tagNew - 1u, val, ent, nl, nuFast, r, abortFlag, xu, vu)) return;
            }
            else
            {
                if (!gs_gather_old4<0, SLAB>(T, G, X, tagNew - 1u, val, ent, nl, nuFast, r, abortFlag, xu, vu)) return;
                if (nuFast > 4)
                    if (!gs_gather_old4<4, SLAB>(T, G, X, tagNew - 1u, val, ent, nl, nuFast, r, abortFlag, xu, vu)) return;
            }
        }
    }
    if (trc && lane == 0) trc[1] = (unsigned long long)wall_clock64();
    // 2. new values of the lower neighbours (the critical path)
    if (!p2p_accumulate<0, false, SLAB>(acc, G, X, tagNew, T.col, val, nullptr, ent, 0, 1, nl, r, abortFlag, waitEst)) return;
    if (trc && lane == 0) trc[2] = (unsigned long long)wall_clock64();
    // 3. upper part, in face order
    if (nuFast >= 0)
    {
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (q < nuFast) acc -= vu[q] * xu[q];
    }
    else if (puWide)
    {
        for (int q = 0; q < nu; q++) acc -= puLds[q * LDU_WAVE + lane];
    }
    else if (j == 0)
    {
        for (int q = nl; q < nl + nu; q++)
        {
            const long e = ent + (long)q * LDU_WAVE;
            acc -= val[e] * psi[T.col[e] & 0x7fffffff];
        }
    }
    else
    {
        if (!p2p_accumulate<0, false, SLAB>(acc, G, X, tagNew - 1u, T.col, val, nullptr, ent, nl, 1, nu, r, abortFlag, waitEst)) return;
    }
    const double out = ldu_div(acc, d, rd);
    if (j == k - 1) psi[r] = out;
    if (SLAB) granule_store_slab(G, X, r, out, tagNew, exported);
    else granule_store(G, r, out, tagNew);
    if (trc && lane == 0)
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trc[3] = (unsigned long long)wall_clock64(); trc[4] = xcc & 0xf; trc[5] = blockIdx.x;
    }
}

// (per-sweep layouts, ldu_gslayouts.cpp: tables and coefficients of sweeps 1 .. 3; a task's slice index counts in ITS sweep's layout)
struct GsLays { SliceTab t[3]; const double* val[3]; int on; };

template <bool LAY>
__device__ __forceinline__ void
sweep_p2p_gs_multi_body(const SliceTab& T, const int* __restrict__ tasks, int nTasks, int nChunks, int k,
                        unsigned* ticket, unsigned ticketBase, int window, unsigned doneBase, uint4* G, unsigned tag0,
                        int* abortFlag, double* psi, const double* rhs, const double* diag, const double* val,
                        int puSlots, const GsLays& L)
{
    __shared__ int s_chunk[2];
    extern __shared__ double s_pu[];   // [wave][puSlots][64]: parked products of wide upper parts
    unsigned* const done = ticket + 32;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    P2PStat waitEst = {0, 0, 0, -1, nullptr};
    int nextT = 0;
    if (threadIdx.x == 0) nextT = (int)(atomicAdd(ticket, 1u) - ticketBase);
    for (int it = 0;; it++)
    {
        if (threadIdx.x == 0)
        {
            if (window && it) atomicAdd(done, 1u);
            const int t = ldu_abort_seen(abortFlag, it) ? 0x7fffffff : nextT;
            if (window && t < nChunks) p2p_window_wait(done, doneBase, t, window, abortFlag);
            s_chunk[it & 1] = t;
            if (t < nChunks) nextT = (int)(atomicAdd(ticket, 1u) - ticketBase);
        }
        __syncthreads();
        const int chunk = s_chunk[it & 1];
        if (chunk >= nChunks) return;
        const int ti = chunk * P2P_CHUNK + wave;
        if (ti < nTasks)
        {
            const int task = __builtin_amdgcn_readfirstlane(tasks[ti]);    // (wave-uniform: ti depends on the wave only)
            if (task >= 0)
            {
                const int sl = task & 0x0fffffff, j = task >> 28;
                if (LAY && j > 0)
                {
                    const SliceTab& Tj = j == 1 ? L.t[0] : (j == 2 ? L.t[1] : L.t[2]);
                    const double* vj = j == 1 ? L.val[0] : (j == 2 ? L.val[1] : L.val[2]);
                    p2p_gs_task<false>(Tj, sl, j, k, lane, G, nullptr, nullptr, tag0, abortFlag, psi, rhs, diag, vj, waitEst,
                                       s_pu + (size_t)wave * puSlots * LDU_WAVE, puSlots);
                }
                else
                    p2p_gs_task<false>(T, sl, j, k, lane, G, nullptr, nullptr, tag0, abortFlag, psi, rhs, diag, val, waitEst,
                                       s_pu + (size_t)wave * puSlots * LDU_WAVE, puSlots);
            }
        }
        if (window) __syncthreads();
    }
}


__global__ void __launch_bounds__(P2P_BLK)
sweep_p2p_gs_multi_kernel(SliceTab T, const int* __restrict__ tasks, int nTasks, int nChunks, int k,
                          unsigned* ticket, unsigned ticketBase, int window, unsigned doneBase, uint4* G, unsigned tag0,
                          int* abortFlag, double* psi, const double* rhs, const double* diag, const double* val, int puSlots)
{
    GsLays none = GsLays();
    sweep_p2p_gs_multi_body<false>(T, tasks, nTasks, nChunks, k, ticket, ticketBase, window, doneBase, G, tag0, abortFlag, psi, rhs, diag,
                                   val, puSlots, none);
}
// ... with per-sweep layouts (ldu_gslayouts.cpp; off by default): three more sets of tables in the argument list
__global__ void __launch_bounds__(P2P_BLK)
sweep_p2p_gs_multi_lay_kernel(SliceTab T, const int* __restrict__ tasks, int nTasks, int nChunks, int k,
                              unsigned* ticket, unsigned ticketBase, int window, unsigned doneBase, uint4* G, unsigned tag0,
                              int* abortFlag, double* psi, const double* rhs, const double* diag, const double* val, int puSlots,
                              GsLays L)
{
    sweep_p2p_gs_multi_body<true>(T, tasks, nTasks, nChunks, k, ticket, ticketBase, window, doneBase, G, tag0, abortFlag, psi, rhs, diag,
                                  val, puSlots, L);
}


// (slab engine twin of sweep_p2p_gs_multi_kernel: per-slab task queues)
__global__ void __launch_bounds__(P2P_BLK)
sweep_slab_gs_multi_kernel(SliceTab T, SlabCtl C, int k, uint4* G, unsigned tag0, int* abortFlag, double* psi,
                           const double* rhs, const double* diag, const double* val, int puSlots)
{
    __shared__ int s_chunk[2];
    extern __shared__ double s_pu[];   // [wave][puSlots][64]
    const int slab = xcc_id();
    if (slab >= C.nSlabs) return;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int first = C.start[slab];
    const int nT = C.start[slab + 1] - first;
    const int nChunks = (nT + P2P_CHUNK - 1) / P2P_CHUNK;
    unsigned* ticket = C.tick + slab;
    P2PStat waitEst = {0, 0, 0, -1, nullptr};
    int nextT = 0;
    const int window = C.window[slab];
    unsigned* const done = C.done + slab;
    if (threadIdx.x == 0)
    {
        __hip_atomic_store(C.tickNext + slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(C.doneNext + slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nextT = (int)atomicAdd(ticket, 1u);
    }
    for (int it = 0;; it++)
    {
        if (threadIdx.x == 0)
        {
            if (window && it) atomicAdd(done, 1u);
            const int t = ldu_abort_seen(abortFlag, it) ? 0x7fffffff : nextT;
            if (window && t < nChunks) p2p_window_wait(done, 0u, t, window, abortFlag);
            s_chunk[it & 1] = t;
            if (t < nChunks) nextT = (int)atomicAdd(ticket, 1u);
        }
        __syncthreads();
        const int chunk = s_chunk[it & 1];
        if (chunk >= nChunks) return;
        const int ti = chunk * P2P_CHUNK + wave;
        if (ti < nT)
        {
            const int task = C.list[first + ti];
            p2p_gs_task<true>(T, task & 0x0fffffff, task >> 28, k, lane, G, C.X, C.xflag, tag0, abortFlag, psi,
                              rhs, diag, val, waitEst, s_pu + (size_t)wave * puSlots * LDU_WAVE, puSlots);
        }
        if (window) __syncthreads();
    }
}

// ---------------------------------------------------------------- small matrices: one wavefront, x in LDS
// The coarse half of a GAMG hierarchy is a chain of small matrices (36 ... 5 k cells) with 15-90
// dependency levels of mostly ONE slice each: a cross-CU hand-off per level (1-1.5 us) is most of
// their smoothing time.  Here ONE wavefront keeps the solution vector in LDS and walks the slices
// of k GaussSeidel sweeps in level order - slices of one level are independent, consecutive levels
// are ordered by the wave's own program order, so there is no barrier and no hand-off at all.
// The coefficients of the next two slices are in flight while a slice is computed (they do not
// depend on the sweep).  Arithmetic per row = GaussSeidelSmoother.C:151-176, identical to the
// other engines (bit-exact).
// Measured (216^3 GAMG hierarchy, 4 sweeps; chip-wide engines in brackets): 36 cells 0.049 ms (0.095), 150 cells
// 0.084 (0.140), 302 cells 0.102 (0.137), 608 cells 0.120 (0.182), 1220 cells 0.186 (0.236), 2454 cells 0.179
// (0.247), 4908 cells 0.319 (0.265): ~0.6-0.7 us per slice, so it is used up to ctx->smallMaxCells (3000) cells.
// (Until the accumulation was written branch-free - see SMALL_STEP - it cost 0.8-1.1 us per slice and lost
//  above ~200 cells.)
// (A 512/1024-thread version with a workgroup barrier per level measured 1.2-2.3 us per level: idle
//  waves either issue the same loads - the CU's address pipeline becomes the bound - or skip them
//  behind a branch, after which the compiler must drain all prefetches at the join.)
#define SMALL_MAX_CELLS 16384
#define SMALL_MAX_LDS (144 * 1024)

template <int W> struct SmallRow { int r; unsigned char nl, nu; int c[W]; double v[W]; double b, d; };

// Branch-free on purpose: the wave always issues the same 4 + 2W loads (clamped to row 0 past the end),
// so the compiler knows exactly how many loads are in flight and never drains them at a join.
template <int W>
__device__ __forceinline__ void small_load(const int* __restrict__ sRow, const int* __restrict__ sEnt,
                                           const SliceTab& T, bool on, int s0, int lane,
                                           const double* __restrict__ rhs, const double* __restrict__ diag,
                                           const double* __restrict__ val, SmallRow<W>& R)
{
    const int s = on ? s0 : 0;
    const int r0 = sRow[s];
    const bool have = on && lane < sRow[s + 1] - r0;
    const int r = have ? r0 + lane : 0;
    const long ent = have ? (long)sEnt[s] + lane : 0;
    R.nl = T.nL[r];   // (no arithmetic on loaded values here: it would wait for them)
    R.nu = T.nU[r];
    R.b = rhs[r];
    R.d = diag[r];
#pragma unroll
    for (int q = 0; q < W; q++)
    {
        const long e = ent + (long)q * LDU_WAVE;   // the entry arrays are padded: always readable
        R.c[q] = T.col[e];
        R.v[q] = val[e];
    }
    R.r = have ? r : -1;
}

template <int W>
__global__ void __launch_bounds__(LDU_WAVE)
gs_small_kernel(SliceTab T, int nSlices, int nCells, int k, double* __restrict__ psi,
                const double* __restrict__ rhs, const double* __restrict__ diag, const double* __restrict__ val)
{
    extern __shared__ double smem[];
    double* x = smem;
    int* sRow = (int*)(x + nCells);
    int* sEnt = sRow + nSlices + 1;
    const int lane = threadIdx.x;
    for (int i = lane; i < nCells; i += LDU_WAVE) x[i] = psi[i];
    for (int i = lane; i <= nSlices; i += LDU_WAVE) sRow[i] = T.sliceRow[i];
    for (int i = lane; i < nSlices; i += LDU_WAVE) sEnt[i] = T.sliceEnt[i];
    __syncthreads();
    // three register sets rotate statically (item i lives in set i % 3), nothing moves them
    SmallRow<W> R0, R1, R2;
    int sNext = 0, sweepNext = 0;             // cursor of the next slice to load
    int left = nSlices * k;
#define SMALL_FILL(R)                                                                     \
    do {                                                                                  \
        small_load<W>(sRow, sEnt, T, sweepNext < k, sNext, lane, rhs, diag, val, (R));    \
        if (++sNext == nSlices) { sNext = 0; sweepNext++; }                               \
    } while (0)
#define SMALL_STEP(CUR, FILL)                                                             \
    do {                                                                                  \
        SMALL_FILL(FILL);                                                                 \
        {                                                                                 \
            double acc = (CUR).b;                                                         \
            const int nn = (int)(CUR).nl + (int)(CUR).nu;                                 \
            /* the denominator's half of the division runs under the LDS reads */         \
            const double rd_ = ldu_div_prepare((CUR).d);                                  \
            /* all W LDS reads in flight, then the products, then ONE dependent subtraction per entry;  \
               entries beyond the row subtract +0.0 (identity for every acc, also -0.0).  Written as   \
               `if (q < nn) acc -= v*x[c]` the compiler sinks each read into its branch: W serial      \
               read-wait-multiply-subtract round trips (ISA checked; 1.4-1.5x slower) */               \
            double xv[W], pr[W];                                                          \
            _Pragma("unroll") for (int q = 0; q < W; q++)                                 \
                xv[q] = x[(CUR).c[q] & (SMALL_MAX_CELLS - 1)];                            \
            _Pragma("unroll") for (int q = 0; q < W; q++) asm volatile("" : "+v"(xv[q])); \
            _Pragma("unroll") for (int q = 0; q < W; q++)                                 \
                pr[q] = q < nn ? (CUR).v[q] * xv[q] : 0.0;                                \
            _Pragma("unroll") for (int q = 0; q < W; q++) acc -= pr[q];                   \
            if ((CUR).r >= 0) x[(CUR).r] = ldu_div(acc, (CUR).d, rd_);                                 \
        }                                                                                 \
        /* the next slice may read what this one wrote: LDS is in order within a wave */  \
        LDU_STEP_FENCE();                                \
        --left;                                                                           \
    } while (0)
    SMALL_FILL(R0);
    SMALL_FILL(R1);
    while (left > 0)
    {
        SMALL_STEP(R0, R2);
        if (left == 0) break;
        SMALL_STEP(R1, R0);
        if (left == 0) break;
        SMALL_STEP(R2, R1);
    }
#undef SMALL_STEP
#undef SMALL_FILL
    for (int i = lane; i < nCells; i += LDU_WAVE) psi[i] = x[i];
}

// k sweeps PIPELINED inside one workgroup: wavefront j runs sweep j over the slices in order and trails sweep j-1 by
// exactly what the data dependence asks for - sweep j may take a slice once sweep j-1 has finished every slice that
// holds an upper neighbour of its rows (need[s] slices; in the reference's sequential loop those are the values of
// the previous sweep, GaussSeidelSmoother.C:151-176).  ONE solution vector in LDS, updated in place by every sweep:
// a row's lower neighbours already hold this sweep's values (same wavefront, program order), its upper neighbours
// still hold the previous sweep's (sweep j+1 cannot have reached them: it waits for sweep j on THIS row first).
// Progress words in LDS (one per sweep, written after the slice's values are in LDS); wavefront 0 never waits, so
// the workgroup cannot deadlock.  nSlices + (k-1)*lag steps instead of k*nSlices: the coarse GAMG levels
// (36 ... 4908 cells of the 216^3 hierarchy) spend their time in this kernel.
template <int W, int NW>
__global__ void __launch_bounds__(LDU_WAVE * NW)
gs_small_pipe_kernel(SliceTab T, int nSlices, int nCells, const int* __restrict__ need, double* __restrict__ psi,
                     const double* __restrict__ rhs, const double* __restrict__ diag, const double* __restrict__ val)
{
    extern __shared__ double smem[];
    double* x = smem;
    int* sRow = (int*)(x + nCells);
    int* sEnt = sRow + nSlices + 1;
    int* sNeed = sEnt + nSlices;
    unsigned* prog = (unsigned*)(sNeed + nSlices);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < nCells; i += LDU_WAVE * NW) x[i] = psi[i];
    for (int i = tid; i <= nSlices; i += LDU_WAVE * NW) sRow[i] = T.sliceRow[i];
    for (int i = tid; i < nSlices; i += LDU_WAVE * NW) { sEnt[i] = T.sliceEnt[i]; sNeed[i] = need[i]; }
    if (tid < NW) prog[tid] = 0u;
    __syncthreads();
    {
        SmallRow<W> R0, R1, R2;
        int sNext = 0;
        int left = nSlices, sCur = 0;
#define PIPE_FILL(R)                                                                      \
    do {                                                                                  \
        small_load<W>(sRow, sEnt, T, sNext < nSlices, sNext, lane, rhs, diag, val, (R));  \
        ++sNext;                                                                          \
    } while (0)
#define PIPE_STEP(CUR, FILL)                                                              \
    do {                                                                                  \
        PIPE_FILL(FILL);                                                                  \
        /* the denominator's half of the division: before the wait for the previous sweep */ \
        const double rd_ = ldu_div_prepare((CUR).d);                                      \
        if (wave > 0)                                                                     \
        {                                                                                 \
            const unsigned want = (unsigned)sNeed[sCur];                                  \
            while (__hip_atomic_load(prog + wave - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) \
                __builtin_amdgcn_s_sleep(1);                                              \
            LDU_LDS_ACQUIRE();                                                            \
        }                                                                                 \
        {                                                                                 \
            double acc = (CUR).b;                                                         \
            const int nn = (int)(CUR).nl + (int)(CUR).nu;                                 \
            double xv[W], pr[W];                                                          \
            _Pragma("unroll") for (int q = 0; q < W; q++)                                 \
                xv[q] = x[(CUR).c[q] & (SMALL_MAX_CELLS - 1)];                            \
            _Pragma("unroll") for (int q = 0; q < W; q++) asm volatile("" : "+v"(xv[q])); \
            _Pragma("unroll") for (int q = 0; q < W; q++)                                 \
                pr[q] = q < nn ? (CUR).v[q] * xv[q] : 0.0;                                \
            _Pragma("unroll") for (int q = 0; q < W; q++) acc -= pr[q];                   \
            if ((CUR).r >= 0) x[(CUR).r] = ldu_div(acc, (CUR).d, rd_);                                 \
        }                                                                                 \
        ++sCur;                                                                           \
        /* release fence on LDS only: the slice's x[] stores are ordered before the progress word (acquire fence after the  \
           consumer's load).  A release ATOMIC would order the global prefetches too: s_waitcnt vmcnt(0) in every step */ \
        LDU_LDS_RELEASE();                                                                \
        if (lane == 0) __hip_atomic_store(prog + wave, (unsigned)sCur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        --left;                                                                           \
    } while (0)
        PIPE_FILL(R0);
        PIPE_FILL(R1);
        while (left > 0)
        {
            PIPE_STEP(R0, R2);
            if (left == 0) break;
            PIPE_STEP(R1, R0);
            if (left == 0) break;
            PIPE_STEP(R2, R1);
        }
#undef PIPE_STEP
#undef PIPE_FILL
    }
    __syncthreads();
    for (int i = tid; i < nCells; i += LDU_WAVE * NW) psi[i] = x[i];
}

// need[s]: slices of the previous sweep that must be complete before slice s (all levels up to the highest level that
// holds an upper neighbour of any row in the levels up to s's own: a running maximum, so it is monotone)
static int small_need_build(ldu_addr* a)
{
    if (a->d_smallNeed) return 0;
    const int nLev = a->nLevels;
    std::vector<int> M(nLev, 0);
    for (int f = 0; f < a->nFaces; f++)
    {
        const int Ll = a->level[a->l[f]], Lu = a->level[a->u[f]];
        if (Lu > M[Ll]) M[Ll] = Lu;
    }
    for (int L = 0; L < nLev; L++)
    {
        if (M[L] < L) M[L] = L;
        if (L && M[L] < M[L - 1]) M[L] = M[L - 1];
    }
    std::vector<int> need(a->nSlices > 0 ? a->nSlices : 1, 0);
    double lag = 0;
    for (int L = 0; L < nLev; L++)
        for (int sl = a->levelSliceStart[L]; sl < a->levelSliceStart[L + 1]; sl++)
        {
            need[sl] = a->levelSliceStart[M[L] + 1];
            lag += need[sl] - sl;
        }
    a->smallLag = a->nSlices ? lag / a->nSlices : 0;
    LDU_CHECK_HIP(hipMalloc((void**)&a->d_smallNeed, sizeof(int) * need.size()));
    LDU_CHECK_HIP(hipMemcpy(a->d_smallNeed, need.data(), sizeof(int) * need.size(), hipMemcpyHostToDevice));
    if (getenv("LDU_VERBOSE"))
        fprintf(stderr, "[ldugpu] small pipelined sweeps: %d cells, %d slices, %d levels, a sweep trails the previous one by "
                        "%.1f slices on average\n", a->nCells, a->nSlices, nLev, a->smallLag);
    return 0;
}

template <int W, int NW>
static int launch_gs_small_pipe(ldu_addr* a, const SliceTab& T, size_t lds, double* psi, const double* rhs,
                                const double* diag, const double* val)
{
    static bool attrSetDev[64] = {false};   // (the attribute is per device: one flag per device ordinal)
    bool& attrSet = attrSetDev[a->ctx->device & 63];
    if (!attrSet)
    {
        LDU_CHECK_HIP(hipFuncSetAttribute((const void*)gs_small_pipe_kernel<W, NW>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, SMALL_MAX_LDS));
        attrSet = true;
    }
    gs_small_pipe_kernel<W, NW><<<1, LDU_WAVE * NW, lds, a->ctx->stream>>>(T, a->nSlices, a->nCells, a->d_smallNeed, psi,
                                                                         rhs, diag, val);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int W>
static int launch_gs_small_pipe_k(ldu_addr* a, const SliceTab& T, size_t lds, int k, double* psi, const double* rhs,
                                  const double* diag, const double* val)
{
    switch (k)
    {
    case 2: return launch_gs_small_pipe<W, 2>(a, T, lds, psi, rhs, diag, val);
    case 3: return launch_gs_small_pipe<W, 3>(a, T, lds, psi, rhs, diag, val);
    case 4: return launch_gs_small_pipe<W, 4>(a, T, lds, psi, rhs, diag, val);
    }
    return -1;
}

template <int W>
static int launch_gs_small(ldu_addr* a, const SliceTab& T, size_t lds, int k, double* psi, const double* rhs,
                           const double* diag, const double* val)
{
    static bool attrSetDev[64] = {false};   // (the attribute is per device: one flag per device ordinal)
    bool& attrSet = attrSetDev[a->ctx->device & 63];
    if (!attrSet)
    {
        LDU_CHECK_HIP(hipFuncSetAttribute((const void*)gs_small_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          SMALL_MAX_LDS));
        attrSet = true;
    }
    gs_small_kernel<W><<<1, LDU_WAVE, lds, a->ctx->stream>>>(T, a->nSlices, a->nCells, k, psi, rhs, diag, val);
    return 0;
}

// returns 1 when the addressing does not qualify
int k_sweep_gs_small(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val)
{
    ldu_ctx* ctx = a->ctx;
    if (!ctx->smallKernels || a->nCells > ctx->smallMaxCells || a->nCells == 0 || k <= 0 || a->maxRowWidth > 16)
        return 1;
    const size_t lds = sizeof(double) * (size_t)a->nCells + sizeof(int) * (3 * ((size_t)a->nSlices + 1)) + 64;
    if (lds > SMALL_MAX_LDS) return 1;
    SliceTab T{a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_nL, a->d_nU, a->d_col};
    const bool pipe = ctx->smallPipe && k >= 2;
    // one wavefront walking every slice of every sweep: only worth it below ~3000 cells; the pipelined form
    // (k wavefronts) carries the larger small levels too
    if (!pipe && a->nCells > 3000) return 1;
    if (pipe && small_need_build(a)) return -1;
    ctx->profStart(a, 4);
    int rc = 0;
    if (pipe)
    {
        int left = k;
        while (left > 0 && !rc)
        {
            const int kk = left > 4 ? (left == 5 ? 3 : 4) : left;    // never leave a single sweep over
            if (kk == 1) break;
            if (a->maxRowWidth <= 8) rc = launch_gs_small_pipe_k<8>(a, T, lds, kk, psi, rhs, diag, val);
            else if (a->maxRowWidth <= 12) rc = launch_gs_small_pipe_k<12>(a, T, lds, kk, psi, rhs, diag, val);
            else rc = launch_gs_small_pipe_k<16>(a, T, lds, kk, psi, rhs, diag, val);
            left -= kk;
        }
        k = left;
    }
    if (!rc && k > 0)
    {
        if (a->maxRowWidth <= 8) rc = launch_gs_small<8>(a, T, lds, k, psi, rhs, diag, val);
        else if (a->maxRowWidth <= 12) rc = launch_gs_small<12>(a, T, lds, k, psi, rhs, diag, val);
        else rc = launch_gs_small<16>(a, T, lds, k, psi, rhs, diag, val);
    }
    ctx->profStop(a, 4);
    if (rc) return -1;
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// Times of the (sweep, slice) tasks of k pipelined GaussSeidel sweeps in their slice-level dependency DAG:
//   T(0, s) = dependency level of s;   T(j, s) = 1 + max( T(j, slices holding lower neighbours of s's rows),
//                                                      T(j-1, slices holding their upper neighbours), T(j-1, s) )
// Sorting the tasks by T gives a topological order in which a task appears as early as its slice can run.
static int gs_slice_dag_times(ldu_addr* a, int k, std::vector<std::vector<int>>& T, int& maxT)
{
    const int nLev = a->nLevels;
    const int nS = a->nSlices;
    std::vector<int> rowSlice(a->nCells);
    {
        std::vector<int> sliceRow(nS + 1);
        LDU_CHECK_HIP(hipMemcpy(sliceRow.data(), a->d_sliceRow, sizeof(int) * (size_t)(nS + 1), hipMemcpyDeviceToHost));
        for (int sl = 0; sl < nS; sl++)
            for (int r = sliceRow[sl]; r < sliceRow[sl + 1]; r++) rowSlice[r] = sl;
    }
    T.assign(k, std::vector<int>(nS, 0));
    for (int L = 0; L < nLev; L++)
        for (int sl = a->levelSliceStart[L]; sl < a->levelSliceStart[L + 1]; sl++) T[0][sl] = L;
    maxT = nLev - 1;
    for (int j = 1; j < k; j++)
    {
        std::vector<int>& Tj = T[j];
        const std::vector<int>& Tp = T[j - 1];
        for (int sl = 0; sl < nS; sl++) Tj[sl] = Tp[sl];
        for (int f = 0; f < a->nFaces; f++)   // previous sweep's values of the upper neighbours
        {
            const int sl = rowSlice[a->iperm[a->l[f]]], su = rowSlice[a->iperm[a->u[f]]];
            if (Tp[su] > Tj[sl]) Tj[sl] = Tp[su];
        }
        for (int sl = 0; sl < nS; sl++) Tj[sl]++;
        // this sweep's values of the lower neighbours: rows in level order (a slice never straddles a level, so
        // every slice is final before a slice of a higher level reads it)
        for (int r = 0; r < a->nCells; r++)
        {
            const int c = a->perm[r], sr = rowSlice[r];
            int t = Tj[sr];
            for (int q = a->losortStart[c]; q < a->losortStart[c + 1]; q++)
            {
                const int tl = Tj[rowSlice[a->iperm[a->l[a->losort[q]]]]] + 1;
                if (tl > t) t = tl;
            }
            Tj[sr] = t;
        }
        for (int sl = 0; sl < nS; sl++) maxT = std::max(maxT, Tj[sl]);
    }
    return 0;
}

// ---------------------------------------------------------------- one-workgroup engine (solution vector in LDS)
// The GAMG levels below ~6 000 cells are all dependency depth and no width: 20 ... 90 levels of 1 ... 2 slices.  The
// single-wavefront kernels above walk the slices one after the other and let sweep j+1 trail sweep j by the worst
// upper-neighbour reach of the whole level (agglomerated levels: half the depth); the chip-wide / slab engines hand over
// through L2 / memory (2-3 us per level).  Here ONE workgroup of NW wavefronts holds the solution vector AND a sweep
// stamp per row in LDS (9 bytes per cell of 160 KB) and runs the k sweeps of a smoothing call as (sweep, slice) tasks
// dealt round-robin to its wavefronts in the order of their time in the slice-level dependency DAG
// (gs_slice_dag_times).  A row of sweep j is ready when its lower neighbours carry stamp j+1 and its upper neighbours
// stamp j - the exact dependencies of the reference's sequential loop (GaussSeidelSmoother.C:147-176), row by row: a
// lane adds up the stamps of its neighbours (a stamp can never be AHEAD of what the row needs, so the sum reaches
// W*j + nLower exactly when every neighbour is ready; entries past the row's end point at the row itself, stamp j).
// The value is written before the stamp and read after it (LDS is in order within a wave).  Global loads of a task
// (its record one step, its rows two steps ahead) are in flight while other wavefronts compute.  Arithmetic per row
// exactly as gs_small_kernel.  Rows of any width: 16 entries from registers, a wider slice's tail in chunks of 8.
#define WG_MAX_LDS (160 * 1024)
struct WgRow { int r, nl, nn, W, j; long ent; int c[16]; double v[16]; double b, d; };

__device__ __forceinline__ void wg_rec_load(const int4* __restrict__ tasks, int nTasks, int i, int4& Q)
{
    Q = tasks[i < nTasks ? i : nTasks];    // (one idle record past the end)
}

__device__ __forceinline__ void wg_row_load(const int4& Q, int lane, const unsigned char* __restrict__ nLv,
                                            const unsigned char* __restrict__ nUv, const int* __restrict__ col,
                                            const double* __restrict__ rhs, const double* __restrict__ diag,
                                            const double* __restrict__ val, WgRow& R)
{
    const int r0 = Q.x, cnt = Q.y & 255;
    const bool have = lane < cnt;
    const int r = have ? r0 + lane : 0;
    const long ent = (long)Q.z + (have ? lane : 0);
    R.nl = nLv[r];
    R.nn = nUv[r];    // (+ nl at the point of use: no arithmetic on loaded values here, it would wait for them)
    R.b = rhs[r];
    R.d = diag[r];
#pragma unroll
    for (int q = 0; q < 16; q++)
    {
        const long e = ent + (long)q * LDU_WAVE;   // the entry arrays are padded: always readable
        R.c[q] = col[e];
        R.v[q] = val[e];
    }
    R.r = have ? r : -1;
    R.W = Q.y >> 8;
    R.j = Q.w;
    R.ent = ent;
}

template <int NW, bool TRACE>
__global__ void __launch_bounds__(LDU_WAVE * NW)
gs_wg_kernel(const unsigned char* __restrict__ nLv, const unsigned char* __restrict__ nUv, const int* __restrict__ col,
             const int4* __restrict__ tasks, int nTasks, int nCells, int* abortFlag, double* __restrict__ psi,
             const double* __restrict__ rhs, const double* __restrict__ diag, const double* __restrict__ val)
{
    extern __shared__ double smem[];
    double* x = smem;
    unsigned char* stamp = (unsigned char*)(x + nCells);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // the first records and rows are on their way while the vector moves into LDS
    WgRow R0, R1, R2;
    int4 Q;
    int iNext = wave;
    wg_rec_load(tasks, nTasks, iNext, Q); iNext += NW;
#define WG_FILL(R)                                                                        \
    do {                                                                                  \
        wg_row_load(Q, lane, nLv, nUv, col, rhs, diag, val, (R));                         \
        wg_rec_load(tasks, nTasks, iNext, Q);                                             \
        iNext += NW;                                                                      \
    } while (0)
    WG_FILL(R0);
    WG_FILL(R1);
    for (int i = tid; i < nCells; i += LDU_WAVE * NW) { x[i] = psi[i]; stamp[i] = 0; }
    __syncthreads();
    bool alive = true;
    int left = (nTasks - wave + NW - 1) / NW;      // this wavefront's tasks
    // (debug timeline, ldu_debug_gs_multi_trace: per task 8 x u64 = step start, loads issued, dependencies seen, stored
    //  [100 MHz wall clock], wavefront, sweep)
    unsigned long long* trc = TRACE && g_gsm_trace ? g_gsm_trace + (size_t)wave * 8 : nullptr;
#define WG_TRC(k) do { if (TRACE && trc && lane == 0) trc[k] = wall_clock64(); } while (0)
    // wait until the stamps of the N entries cc[] add up to `want` (every neighbour ready), then fetch their values
#define WG_WAIT(N, cc, want, xv)                                                          \
    do {                                                                                  \
        unsigned spins = 0;                                                               \
        unsigned long long tw0 = 0;                                                       \
        while (alive)                                                                     \
        {                                                                                 \
            int st[N];                                                                    \
            _Pragma("unroll") for (int q = 0; q < N; q++) st[q] = stamp[cc[q]];           \
            int sum = 0;                                                                  \
            _Pragma("unroll") for (int q = 0; q < N; q++) sum += st[q];                   \
            if (__builtin_amdgcn_ballot_w64(have && sum != (want)) == 0ull) break;        \
            /* (sixteen LDS reads per round already pace this loop; tighter polling - one watched byte, values fetched \
               with the stamps - took the LDS and the issue slots from the wavefront everybody waits for: 2x slower) */ \
            if (ldu_wait_expired(spins, 1u << 30, abortFlag, tw0)) { *abortFlag = 1; alive = false; } \
        }                                                                                 \
        LDU_LDS_ACQUIRE();                                                                \
        _Pragma("unroll") for (int q = 0; q < N; q++) xv[q] = x[cc[q]];                   \
    } while (0)
#define WG_STEP(CUR, FILL)                                                                \
    do {                                                                                  \
        WG_TRC(0);                                                                        \
        WG_FILL(FILL);                                                                    \
        WG_TRC(1);                                                                        \
        const double rd_ = ldu_div_prepare((CUR).d);                                      \
        {                                                                                 \
            const bool have = (CUR).r >= 0;                                               \
            const int self = have ? (CUR).r : 0;                                          \
            const int nl = (CUR).nl, nn = have ? nl + (CUR).nn : 0, j = (CUR).j;          \
            int cc[16];                                                                   \
            _Pragma("unroll") for (int q = 0; q < 16; q++) cc[q] = q < nn ? (CUR).c[q] : self; \
            /* (a lane without a row reads stamp 0 sixteen times over and does not vote) */ \
            const int want = 16 * j + (nl < 16 ? nl : 16);                                \
            double acc = (CUR).b;                                                         \
            double xv[16], pr[16];                                                        \
            WG_WAIT(16, cc, want, xv);                                                    \
            WG_TRC(2);                                                                    \
            _Pragma("unroll") for (int q = 0; q < 16; q++) asm volatile("" : "+v"(xv[q])); \
            _Pragma("unroll") for (int q = 0; q < 16; q++)                                \
                pr[q] = q < nn ? (CUR).v[q] * xv[q] : 0.0;                                \
            _Pragma("unroll") for (int q = 0; q < 16; q++) acc -= pr[q];                  \
            if ((CUR).W > 16)                                                             \
                for (int q0 = 16; q0 < (CUR).W; q0 += 8)                                  \
                {                                                                         \
                    int c8[8]; double v8[8];                                              \
                    _Pragma("unroll") for (int q = 0; q < 8; q++)                         \
                    {                                                                     \
                        const long e = (CUR).ent + (long)(q0 + q) * LDU_WAVE;             \
                        c8[q] = col[e]; v8[q] = val[e];                                   \
                    }                                                                     \
                    _Pragma("unroll") for (int q = 0; q < 8; q++) c8[q] = q0 + q < nn ? c8[q] : self; \
                    const int lo = nl - q0 < 0 ? 0 : (nl - q0 > 8 ? 8 : nl - q0);         \
                    const int want8 = 8 * j + lo;                                         \
                    WG_WAIT(8, c8, want8, xv);                                            \
                    _Pragma("unroll") for (int q = 0; q < 8; q++)                         \
                        pr[q] = q0 + q < nn ? v8[q] * xv[q] : 0.0;                        \
                    _Pragma("unroll") for (int q = 0; q < 8; q++) acc -= pr[q];           \
                }                                                                         \
            if (have)                                                                     \
            {                                                                             \
                x[(CUR).r] = ldu_div(acc, (CUR).d, rd_);                                  \
                /* release (LDS only): the value is in LDS before its stamp moves */      \
                LDU_LDS_RELEASE();                                                        \
                stamp[(CUR).r] = (unsigned char)(j + 1);                                  \
            }                                                                             \
        }                                                                                 \
        LDU_STEP_FENCE();                                                                 \
        if (TRACE && trc && lane == 0) { trc[3] = wall_clock64(); trc[4] = wave; trc[5] = (CUR).j; trc += (size_t)NW * 8; } \
        --left;                                                                           \
    } while (0)
    while (left > 0)
    {
        WG_STEP(R0, R2);
        if (left == 0) break;
        WG_STEP(R1, R0);
        if (left == 0) break;
        WG_STEP(R2, R1);
    }
#undef WG_STEP
#undef WG_WAIT
#undef WG_FILL
#undef WG_TRC
    __syncthreads();
    for (int i = tid; i < nCells; i += LDU_WAVE * NW) psi[i] = x[i];
}

// GaussSeidel with processor patches on a small level, ALL k sweeps in one launch (peer-store backend): what the launch
// needs to exchange the boundary values of every sweep itself (kernel-private window regions of the addressing, PeerHalo)
struct WgPeer {
    uint4* const* kdst;            // [2][nPF] (null: no remote faces)
    const uint4* const* ksrc;
    unsigned* kseq;
    int nPF;
    const int* pfCell;             // [nPF] row of the face's cell (plan numbering)
    const int* cycPair;            // [nPF] paired face of a cyclic face, -1 = remote
    const double* bou;             // [nPF] interfaceBouCoeffs
    int nBRows;
    const int* bRow; const int* bStart; const int* bFace;
    const int* rowB;               // [nCells] row -> boundary row (-1)
    int kSweeps;
};

template <int NW>
__global__ void __launch_bounds__(LDU_WAVE * NW)
gs_wg_peer_kernel(WgPeer PS, const unsigned char* __restrict__ nLv, const unsigned char* __restrict__ nUv, const int* __restrict__ col,
             const int4* __restrict__ tasks, int nTasks, int nCells, int* abortFlag, double* __restrict__ psi,
             const double* __restrict__ rhs, const double* __restrict__ diag, const double* __restrict__ val)
{
    extern __shared__ double smem[];
    double* x = smem;
    unsigned char* stamp = (unsigned char*)(x + nCells);
    double* bP = (double*)(stamp + ((nCells + 7) & ~7));      // [nBRows] this sweep's bPrime of the boundary rows
    constexpr bool TRACE = false;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // the first records and rows are on their way while the vector moves into LDS
    WgRow R0, R1, R2;
    int4 Q;
    int iNext = wave;
#define WG_FILL(R)                                                                        \
    do {                                                                                  \
        wg_row_load(Q, lane, nLv, nUv, col, rhs, diag, val, (R));                         \
        (R).j = sweep;                                                                    \
        { const int sb_ = (R).r >= 0 ? PS.rowB[(R).r] : -1; if (sb_ >= 0) (R).b = bP[sb_]; } \
        wg_rec_load(tasks, nTasks, iNext, Q);                                             \
        iNext += NW;                                                                      \
    } while (0)
    for (int i = tid; i < nCells; i += LDU_WAVE * NW) { x[i] = psi[i]; stamp[i] = 0; }
    unsigned hseq = PS.kseq ? *PS.kseq : 0u;
    __syncthreads();
    bool alive = true;
    int left = 0;
    for (int sweep = 0; sweep < PS.kSweeps; sweep++)
    {
    // ---- GaussSeidelSmoother.C:98-145 for this sweep: bPrime = source, the coupled boundaries Jacobi-style with negated
    // coefficients, the neighbour values being the OTHER rank's psi after its previous sweep - stored into its window by
    // that rank's workgroup right here (peer_store), polled from this rank's window (ldu_peer.hip's protocol, the
    // sequence number kept on the device because the whole k-sweep smoothing is one launch)
    {
        ++hseq;
        const size_t par = (size_t)(hseq & 1u) * PS.nPF;
        for (int i = tid; i < PS.nPF; i += LDU_WAVE * NW)
        {
            uint4* d = PS.kdst ? PS.kdst[par + i] : nullptr;
            if (d) peer_store(d, x[PS.pfCell[i]], hseq);
        }
        for (int jb = tid; jb < PS.nBRows; jb += LDU_WAVE * NW)
        {
            double acc = rhs[PS.bRow[jb]];
            for (int t = PS.bStart[jb]; t < PS.bStart[jb + 1]; t++)
            {
                const int i = PS.bFace[t];
                const int cp = PS.cycPair[i];
                double pn = 0.0;
                if (cp >= 0) pn = x[PS.pfCell[cp]];
                else
                {
                    const uint4* sp = PS.ksrc[par + i];
                    unsigned spins = 0;
                    unsigned long long tw0 = 0;
                    while (!peer_load(sp, hseq, pn))
                    {
                        if (peer_wait_expired(spins, tw0, abortFlag)) { abortFlag[LDU_PEER_FLAG] = 1; pn = 0.0; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const double c = -PS.bou[i];
                acc -= c * pn;
            }
            bP[jb] = acc;
        }
        __syncthreads();
    }
    iNext = wave;
    wg_rec_load(tasks, nTasks, iNext, Q); iNext += NW;
    WG_FILL(R0);
    WG_FILL(R1);
    left = (nTasks - wave + NW - 1) / NW;      // this wavefront's tasks
    // (debug timeline, ldu_debug_gs_multi_trace: per task 8 x u64 = step start, loads issued, dependencies seen, stored
    //  [100 MHz wall clock], wavefront, sweep)
    unsigned long long* trc = TRACE && g_gsm_trace ? g_gsm_trace + (size_t)wave * 8 : nullptr;
#define WG_TRC(k) do { if (TRACE && trc && lane == 0) trc[k] = wall_clock64(); } while (0)
    // wait until the stamps of the N entries cc[] add up to `want` (every neighbour ready), then fetch their values
#define WG_WAIT(N, cc, want, xv)                                                          \
    do {                                                                                  \
        unsigned spins = 0;                                                               \
        unsigned long long tw0 = 0;                                                       \
        while (alive)                                                                     \
        {                                                                                 \
            int st[N];                                                                    \
            _Pragma("unroll") for (int q = 0; q < N; q++) st[q] = stamp[cc[q]];           \
            int sum = 0;                                                                  \
            _Pragma("unroll") for (int q = 0; q < N; q++) sum += st[q];                   \
            if (__builtin_amdgcn_ballot_w64(have && sum != (want)) == 0ull) break;        \
            /* (sixteen LDS reads per round already pace this loop; tighter polling - one watched byte, values fetched \
               with the stamps - took the LDS and the issue slots from the wavefront everybody waits for: 2x slower) */ \
            if (ldu_wait_expired(spins, 1u << 30, abortFlag, tw0)) { *abortFlag = 1; alive = false; } \
        }                                                                                 \
        LDU_LDS_ACQUIRE();                                                                \
        _Pragma("unroll") for (int q = 0; q < N; q++) xv[q] = x[cc[q]];                   \
    } while (0)
#define WG_STEP(CUR, FILL)                                                                \
    do {                                                                                  \
        WG_TRC(0);                                                                        \
        WG_FILL(FILL);                                                                    \
        WG_TRC(1);                                                                        \
        const double rd_ = ldu_div_prepare((CUR).d);                                      \
        {                                                                                 \
            const bool have = (CUR).r >= 0;                                               \
            const int self = have ? (CUR).r : 0;                                          \
            const int nl = (CUR).nl, nn = have ? nl + (CUR).nn : 0, j = (CUR).j;          \
            int cc[16];                                                                   \
            _Pragma("unroll") for (int q = 0; q < 16; q++) cc[q] = q < nn ? (CUR).c[q] : self; \
            /* (a lane without a row reads stamp 0 sixteen times over and does not vote) */ \
            const int want = 16 * j + (nl < 16 ? nl : 16);                                \
            double acc = (CUR).b;                                                         \
            double xv[16], pr[16];                                                        \
            WG_WAIT(16, cc, want, xv);                                                    \
            WG_TRC(2);                                                                    \
            _Pragma("unroll") for (int q = 0; q < 16; q++) asm volatile("" : "+v"(xv[q])); \
            _Pragma("unroll") for (int q = 0; q < 16; q++)                                \
                pr[q] = q < nn ? (CUR).v[q] * xv[q] : 0.0;                                \
            _Pragma("unroll") for (int q = 0; q < 16; q++) acc -= pr[q];                  \
            if ((CUR).W > 16)                                                             \
                for (int q0 = 16; q0 < (CUR).W; q0 += 8)                                  \
                {                                                                         \
                    int c8[8]; double v8[8];                                              \
                    _Pragma("unroll") for (int q = 0; q < 8; q++)                         \
                    {                                                                     \
                        const long e = (CUR).ent + (long)(q0 + q) * LDU_WAVE;             \
                        c8[q] = col[e]; v8[q] = val[e];                                   \
                    }                                                                     \
                    _Pragma("unroll") for (int q = 0; q < 8; q++) c8[q] = q0 + q < nn ? c8[q] : self; \
                    const int lo = nl - q0 < 0 ? 0 : (nl - q0 > 8 ? 8 : nl - q0);         \
                    const int want8 = 8 * j + lo;                                         \
                    WG_WAIT(8, c8, want8, xv);                                            \
                    _Pragma("unroll") for (int q = 0; q < 8; q++)                         \
                        pr[q] = q0 + q < nn ? v8[q] * xv[q] : 0.0;                        \
                    _Pragma("unroll") for (int q = 0; q < 8; q++) acc -= pr[q];           \
                }                                                                         \
            if (have)                                                                     \
            {                                                                             \
                x[(CUR).r] = ldu_div(acc, (CUR).d, rd_);                                  \
                /* release (LDS only): the value is in LDS before its stamp moves */      \
                LDU_LDS_RELEASE();                                                        \
                stamp[(CUR).r] = (unsigned char)(j + 1);                                  \
            }                                                                             \
        }                                                                                 \
        LDU_STEP_FENCE();                                                                 \
        if (TRACE && trc && lane == 0) { trc[3] = wall_clock64(); trc[4] = wave; trc[5] = (CUR).j; trc += (size_t)NW * 8; } \
        --left;                                                                           \
    } while (0)
    while (left > 0)
    {
        WG_STEP(R0, R2);
        if (left == 0) break;
        WG_STEP(R1, R0);
        if (left == 0) break;
        WG_STEP(R2, R1);
    }
    __syncthreads();      // the sweep is complete in LDS before its boundary values travel / the next sweep's rows load
    }
#undef WG_STEP
#undef WG_WAIT
#undef WG_FILL
#undef WG_TRC
    if (tid == 0 && PS.kseq) *PS.kseq = hseq;
    for (int i = tid; i < nCells; i += LDU_WAVE * NW) psi[i] = x[i];
}


template <int NW>
static int launch_gs_wg(ldu_addr* a, const ldu_addr::WgTasks& W, size_t lds, double* psi, const double* rhs, const double* diag,
                        const double* val)
{
    static bool attrSetDev[64] = {false};   // (the attribute is per device: one flag per device ordinal)
    bool& attrSet = attrSetDev[a->ctx->device & 63];
    if (!attrSet)
    {
        LDU_CHECK_HIP(hipFuncSetAttribute((const void*)gs_wg_kernel<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_MAX_LDS));
        LDU_CHECK_HIP(hipFuncSetAttribute((const void*)gs_wg_kernel<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_MAX_LDS));
        attrSet = true;
    }
    if (g_wg_trace_on)
        gs_wg_kernel<NW, true><<<1, LDU_WAVE * NW, lds, a->ctx->stream>>>(a->d_nL, a->d_nU, a->d_col, (const int4*)W.d_tasks, W.n,
                                                                        a->nCells, a->ctx->d_abort, psi, rhs, diag, val);
    else
        gs_wg_kernel<NW, false><<<1, LDU_WAVE * NW, lds, a->ctx->stream>>>(a->d_nL, a->d_nU, a->d_col, (const int4*)W.d_tasks, W.n,
                                                                         a->nCells, a->ctx->d_abort, psi, rhs, diag, val);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

static inline bool wg_qualifies(const ldu_addr* a)
{
    const ldu_ctx* ctx = a->ctx;
    return ctx->wgEngine && a->wgLevel && a->nCells > 0 && 9 * (size_t)a->nCells + 64 <= WG_MAX_LDS;
}

// k GaussSeidel sweeps (k = 1 ... 4) of a small matrix in one workgroup; returns 1 when the addressing does not qualify
// the (sweep, slice) task list of k sweeps for the one-workgroup engine (cached per k in the addressing)
static int wg_tasks(ldu_addr* a, int k, const ldu_addr::WgTasks** out)
{
    auto it = a->wgTasks.find(k);
    if (it == a->wgTasks.end())
    {
        std::vector<std::vector<int>> T;
        int maxT = 0;
        if (gs_slice_dag_times(a, k, T, maxT)) return -1;
        const int nS = a->nSlices;
        std::vector<int> sliceRow(nS + 1), sliceCnt(nS), sliceEnt(nS), sliceW(nS);
        LDU_CHECK_HIP(hipMemcpy(sliceRow.data(), a->d_sliceRow, sizeof(int) * (size_t)(nS + 1), hipMemcpyDeviceToHost));
        LDU_CHECK_HIP(hipMemcpy(sliceCnt.data(), a->d_sliceCnt, sizeof(int) * (size_t)nS, hipMemcpyDeviceToHost));
        LDU_CHECK_HIP(hipMemcpy(sliceEnt.data(), a->d_sliceEnt, sizeof(int) * (size_t)nS, hipMemcpyDeviceToHost));
        LDU_CHECK_HIP(hipMemcpy(sliceW.data(), a->d_sliceW, sizeof(int) * (size_t)nS, hipMemcpyDeviceToHost));
        // counting sort by T; inside one T the sweeps ascend (the earlier sweep is what everything else waits for)
        std::vector<long> start((size_t)maxT + 2, 0);
        for (int j = 0; j < k; j++)
            for (int sl = 0; sl < nS; sl++) start[(size_t)T[j][sl] + 1]++;
        for (size_t i = 0; i + 1 < start.size(); i++) start[i + 1] += start[i];
        std::vector<int> rec((size_t)4 * ((size_t)k * nS + 1), 0);
        for (int j = 0; j < k; j++)
            for (int sl = 0; sl < nS; sl++)
            {
                int* r4 = rec.data() + 4 * (size_t)start[T[j][sl]]++;
                r4[0] = sliceRow[sl]; r4[1] = sliceCnt[sl] | (sliceW[sl] << 8); r4[2] = sliceEnt[sl];
                r4[3] = j;
            }
        ldu_addr::WgTasks W;
        W.n = k * nS;       // (the record past the end stays zero: no rows)
        W.steps = maxT + 1;
        LDU_CHECK_HIP(hipMalloc((void**)&W.d_tasks, sizeof(int) * rec.size()));
        LDU_CHECK_HIP(hipMemcpy(W.d_tasks, rec.data(), sizeof(int) * rec.size(), hipMemcpyHostToDevice));
        if (getenv("LDU_VERBOSE"))
            fprintf(stderr, "[ldugpu] workgroup engine plan: %d cells, %d slices, %d levels, k = %d: %d steps in the slice-level DAG\n",
                    a->nCells, nS, a->nLevels, k, maxT + 1);
        it = a->wgTasks.emplace(k, W).first;
    }
    *out = &it->second;
    return 0;
}

int k_sweep_gs_wg(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag, const double* val)
{
    ldu_ctx* ctx = a->ctx;
    if (!wg_qualifies(a) || k <= 0 || k > 4) return 1;
    const size_t lds = 9 * (size_t)a->nCells + 64;
    const ldu_addr::WgTasks* W = nullptr;
    if (wg_tasks(a, k, &W)) return -1;
    ctx->profStart(a, 4);
    int rc;
    switch (ctx->wgWaves)
    {
    case 4: rc = launch_gs_wg<4>(a, *W, lds, psi, rhs, diag, val); break;
    default: rc = launch_gs_wg<8>(a, *W, lds, psi, rhs, diag, val); break;
    }
    ctx->profStop(a, 4);
    return rc ? -1 : 0;
}

// GaussSeidel with coupled patches on a small level: k sweeps AND their k boundary exchanges in one launch (peer-store
// backend, kernel-private window regions).  `usable` = every rank of the communicator takes this path for this level
// (decided collectively by the caller: a launch that talks to its neighbours needs all of them).  1 = not taken.
int k_sweep_gs_wg_peer(ldu_addr* a, int k, double* psi, const double* source, const double* diag, const double* val,
                       const double* bou, const int* d_cycPair)
{
    ldu_ctx* ctx = a->ctx;
    // (the sweep stamps of a row are bytes - stamp = sweeps done, want = 16 sweep + lower entries: beyond 15 sweeps per launch the
    //  sums no longer fit; a smoothing call with more sweeps goes sweep by sweep on the caller's path)
    if (!wg_qualifies(a) || k <= 0 || k > 15) return 1;
    const ldu_addr::WgTasks* W = nullptr;
    if (wg_tasks(a, 1, &W)) return -1;

    WgPeer PS = WgPeer();
    if (a->peer) { PS.kdst = a->peer->d_kdst; PS.ksrc = a->peer->d_ksrc; PS.kseq = a->peer->d_kseq; }
    PS.nPF = a->nPatchFaces;
    PS.pfCell = a->d_pfCell;
    PS.cycPair = d_cycPair;
    PS.bou = bou;
    PS.nBRows = a->nBRows;
    PS.bRow = a->d_bRow; PS.bStart = a->d_bStart; PS.bFace = a->d_bFace;
    PS.rowB = a->d_nbRowB;
    PS.kSweeps = k;
    const size_t lds = 9 * (size_t)a->nCells + 64 + 8 * (size_t)a->nBRows + 16;
    if (lds > WG_MAX_LDS) return 1;
    static bool attrSetDev[64] = {false};
    bool& attrSet = attrSetDev[ctx->device & 63];
    if (!attrSet)
    {
        LDU_CHECK_HIP(hipFuncSetAttribute((const void*)gs_wg_peer_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_MAX_LDS));
        attrSet = true;
    }
    ctx->profStart(a, 4);
    gs_wg_peer_kernel<8><<<1, LDU_WAVE * 8, lds, ctx->stream>>>(PS, a->d_nL, a->d_nU, a->d_col, (const int4*)W->d_tasks, W->n,
                                                                a->nCells, ctx->d_abort, psi, source, diag, val);
    LDU_CHECK_HIP(hipGetLastError());
    ctx->profStop(a, 4);
    return 0;
}
bool k_wg_peer_eligible(ldu_addr* a)
{
    PeerKernelComm K;
    if (!comm_peer_kernel_comm(a->ctx, &K) || !wg_qualifies(a) || !a->nPatchFaces) return false;
    bool remote = false;
    for (auto& P : a->patches) if (P.nbrPatch < 0 && P.n) remote = true;
    if (remote && !(a->peer && a->peer->kAll)) return false;
    return 9 * (size_t)a->nCells + 64 + 8 * (size_t)a->nBRows + 16 <= WG_MAX_LDS;
}

int k_set_peer_timeout_kernels(unsigned long long ticks)
{
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_peer_budget), &ticks, sizeof(ticks)));
    return 0;
}

// Host side: topological task order for k pipelined sweeps (cached per k in the addressing; an entry with n < 0 = this
// addressing does not pipeline).  Built at the first smoothing call that needs it - or ahead of it, on the set-up threads of
// a GAMG hierarchy (k_gs_prebuild): on the 12.7 M-cell motorBike mesh the orders of all levels were 2.3 s of the first solve.
static int gs_tasks_ensure(ldu_addr* a, int k)
{
    ldu_ctx* ctx = a->ctx;
    if (a->gsTasks.find(k) != a->gsTasks.end()) return 0;
    {

        // M[L] = running max over levels <= L of the highest level holding an upper neighbour
        const int nLev = a->nLevels;
        std::vector<int> M(nLev, 0);
        for (int f = 0; f < a->nFaces; f++)
        {
            const int Ll = a->level[a->l[f]], Lu = a->level[a->u[f]];
            if (Lu > M[Ll]) M[Ll] = Lu;
        }
        int maxSkew = 0;
        for (int L = 0; L < nLev; L++)
        {
            if (M[L] < L) M[L] = L;
            if (L && M[L] < M[L - 1]) M[L] = M[L - 1];
            if (M[L] - L > maxSkew) maxSkew = M[L] - L;
        }
        if (getenv("LDU_VERBOSE"))
            fprintf(stderr, "[ldugpu] GS pipeline plan: %d cells, %d levels, max upper-neighbour skew %d levels\n",
                    a->nCells, nLev, maxSkew);
        // On irregular DAGs the upper neighbours can sit dozens or hundreds of levels ahead.  Round 1 ran such
        // addressings sweep by sweep (spin-bound aborts at skew ~70: thousands of waves far ahead of the front polled
        // so hard that the front itself crawled); with the run-ahead window (p2p_window_wait) they pipeline like the
        // others: irregular 100^3 graph, 1463 levels, skew 262: 4 sweeps 7.2 ms instead of 14.8.  The limit remains as
        // a knob (LDU_GS_MAXSKEW).
        if (maxSkew > ctx->gsPipelineMaxSkew && a->nSlices > 512)
        {
            ldu_addr::GsTasks none;
            none.n = -1;
            a->gsTasks.emplace(k, none);
            return 0;
        }
        // Task order = (sweep, slice) pairs sorted by their time in the slice-level DAG of the k sweeps:
        //   T(0, s) = dependency level of s;   T(j, s) = 1 + max( T(j, slices holding lower neighbours of s's rows),
        //                                                      T(j-1, slices holding their upper neighbours), T(j-1, s) )
        // - a topological order (T grows along every dependency), in which a task appears as early as its slice can
        // run.  (Until round 3 sweep j+1 took a whole LEVEL once sweep j had passed the running maximum of the levels
        // holding upper neighbours: on irregular graphs one far-reaching row per level serialised the sweeps.)
        std::vector<int> tasks;
        tasks.reserve((size_t)k * a->nSlices);
        {
            const int nS = a->nSlices;
            std::vector<std::vector<int>> T;
            int maxT = 0;
            if (gs_slice_dag_times(a, k, T, maxT)) return -1;
            // counting sort by T; inside one T the sweeps ascend and the slices keep their order
            std::vector<long> start((size_t)maxT + 2, 0);
            for (int j = 0; j < k; j++)
                for (int sl = 0; sl < nS; sl++) start[(size_t)T[j][sl] + 1]++;
            for (size_t i = 0; i + 1 < start.size(); i++) start[i + 1] += start[i];
            tasks.assign((size_t)k * nS, 0);
            for (int j = 0; j < k; j++)
                for (int sl = 0; sl < nS; sl++) tasks[(size_t)start[T[j][sl]]++] = (j << 28) | sl;
            if (getenv("LDU_VERBOSE"))
                fprintf(stderr, "[ldugpu] GS pipeline plan: %d cells, %d slices, k = %d: %d steps in the slice-level DAG "
                                "(one sweep: %d levels; level-granular order: ~%d)\n",
                        a->nCells, nS, k, maxT + 1, nLev, nLev + (k - 1) * maxSkew);
        }
        if (tasks.size() != (size_t)k * (size_t)a->nSlices)
        {
            ldu_set_error("GS pipeline: the task order does not cover every (sweep, slice) pair");
            return -1;
        }
        ldu_addr::GsTasks gt;
        gt.n = (int)tasks.size();
        LDU_CHECK_HIP(hipMalloc((void**)&gt.d_tasks, sizeof(int) * (tasks.size() + 1)));
        LDU_CHECK_HIP(hipMemcpy(gt.d_tasks, tasks.data(), sizeof(int) * tasks.size(), hipMemcpyHostToDevice));
        if (a->nSlabs > 0)
        {
            // per-slab queues = the global topological order restricted to each slab
            std::vector<int> sliceSlab(a->nSlices, 0), slabTasks;
            {
                std::vector<int> list(a->nSlices);
                LDU_CHECK_HIP(hipMemcpy(list.data(), a->d_slabList, sizeof(int) * (size_t)a->nSlices,
                                        hipMemcpyDeviceToHost));
                for (int sl = 0; sl < a->nSlabs; sl++)
                    for (int i = a->slabStart[sl]; i < a->slabStart[sl + 1]; i++) sliceSlab[list[i]] = sl;
            }
            slabTasks.reserve(tasks.size());
            for (int sl = 0; sl < a->nSlabs; sl++)
            {
                gt.slabStart[sl] = (int)slabTasks.size();
                for (int t : tasks)
                    if (sliceSlab[t & 0x0fffffff] == sl) slabTasks.push_back(t);
            }
            for (int sl = a->nSlabs; sl <= 8; sl++) gt.slabStart[sl] = (int)slabTasks.size();
            LDU_CHECK_HIP(hipMalloc((void**)&gt.d_slabTasks, sizeof(int) * (slabTasks.size() + 1)));
            LDU_CHECK_HIP(hipMemcpy(gt.d_slabTasks, slabTasks.data(), sizeof(int) * slabTasks.size(),
                                    hipMemcpyHostToDevice));
        }
        a->gsTasks.emplace(k, gt);
    }
    return 0;
}

// ... and with per-sweep layouts (ldu_gslayouts.cpp): task (j, slice of layout j) at the time T_j of the slice's rows - any order
// by T is topological; cached under key k + 16.  1 = no layouts on this addressing (the caller takes the list above).
bool gs_layouts_wanted(const ldu_addr* a);
int gs_layouts_ensure(ldu_addr* a, int k);
int gs_layout_values(ldu_addr* a, const double* levelVal, int k, hipStream_t s, const double* out[4]);
static int gs_layout_tasks_ensure(ldu_addr* a, int k)
{
    if (g_wg_trace_on || !gs_layouts_wanted(a) || (a->nSlabs > 0 && use_slab(a, 2, k))) return 1;
    if (a->gsTasks.find(k + 16) != a->gsTasks.end()) return a->gsTasks[k + 16].n < 0 ? 1 : 0;
    const int rc = gs_layouts_ensure(a, k);
    if (rc < 0) return -1;
    ldu_addr::GsTasks gt;
    gt.layouts = true;
    if (rc > 0) { gt.n = -1; a->gsTasks.emplace(k + 16, gt); return 1; }
    int maxT = a->nLevels - 1;
    for (int j = 1; j < k; j++)
        for (int t : a->gsLay[j]->sliceTime) maxT = std::max(maxT, t);
    std::vector<long> start((size_t)maxT + 2, 0);
    for (int L = 0; L < a->nLevels; L++) start[(size_t)L + 1] += a->levelSliceStart[L + 1] - a->levelSliceStart[L];
    for (int j = 1; j < k; j++)
        for (int t : a->gsLay[j]->sliceTime) start[(size_t)t + 1]++;
    for (size_t i = 0; i + 1 < start.size(); i++) start[i + 1] += start[i];
    std::vector<int> tasks((size_t)start.back(), 0);
    // inside one T the sweeps ascend (the earlier sweep is what everything else waits for) and the slices keep their order
    for (int L = 0; L < a->nLevels; L++)
        for (int sl = a->levelSliceStart[L]; sl < a->levelSliceStart[L + 1]; sl++) tasks[(size_t)start[L]++] = sl;
    for (int j = 1; j < k; j++)
    {
        const std::vector<int>& st = a->gsLay[j]->sliceTime;
        if (st.size() >= (1u << 28)) { ldu_set_error("GS layouts: too many slices"); return -1; }
        for (size_t sl = 0; sl < st.size(); sl++) tasks[(size_t)start[st[sl]]++] = (j << 28) | (int)sl;
    }
    gt.n = (int)tasks.size();
    LDU_CHECK_HIP(hipMalloc((void**)&gt.d_tasks, sizeof(int) * (tasks.size() + 1)));
    LDU_CHECK_HIP(hipMemcpy(gt.d_tasks, tasks.data(), sizeof(int) * tasks.size(), hipMemcpyHostToDevice));
    if (getenv("LDU_VERBOSE"))
        fprintf(stderr, "[ldugpu] GS pipeline plan with per-sweep layouts: %d cells, k = %d: %d tasks, %d steps in the row-level DAG "
                        "(one sweep: %d levels)\n", a->nCells, k, gt.n, maxT + 1, a->nLevels);
    a->gsTasks.emplace(k + 16, gt);
    return 0;
}

int k_gs_prebuild(ldu_addr* a, int k)
{
    ldu_ctx* ctx = a->ctx;
    if (k < 2 || k > 4 || a->nCells == 0 || a->nPatchFaces || !ctx->sweepP2P || !ctx->gsPipeline) return 0;
    const int e = k_engine_of(a, 2);
    if (e == 6)
    {
        const int rb = k_blocks_prebuild(a, k);
        if (rb < 0) return -1;
        if (rb != 2) return 0;
        const int rl = gs_layout_tasks_ensure(a, k);      // this k stays on the level engines
        if (rl <= 0) return rl;
        return gs_tasks_ensure(a, k);
    }
    if (e != 0 && e != 1) return 0;      // one workgroup / single wavefront / clusters: their own (cheap) plans
    const int rl = gs_layout_tasks_ensure(a, k);
    if (rl <= 0) return rl;
    return gs_tasks_ensure(a, k);
}

int k_sweep_gs_multi(ldu_addr* a, int k, double* psi, const double* rhs, const double* diag,
                     const double* val)
{
    ldu_ctx* ctx = a->ctx;
    ldu_addr::P2PLane& P = *a->lane(0);
    hipStream_t s = ctx->stream;
    if (a->nCells == 0 || k <= 0) return 0;
    // per-sweep layouts (ldu_gslayouts.cpp) where they apply and the coefficients' origin is known
    GsLays LY = GsLays();
    bool lay = false;
    if (k >= 2 && k <= 4)
    {
        const int rl = gs_layout_tasks_ensure(a, k);
        if (rl < 0) return -1;
        if (rl == 0)
        {
            const double* lv[4];
            const int rv = gs_layout_values(a, val, k, s, lv);
            if (rv < 0) return -1;
            if (rv == 0)
            {
                lay = true;
                LY.on = 1;
                for (int j = 1; j < k; j++)
                {
                    const ldu_addr::GsLayout* Y = a->gsLay[j];
                    SliceTab& t = LY.t[j - 1];
                    t.sliceRow = Y->d_sliceRow; t.sliceCnt = Y->d_sliceCnt; t.sliceEnt = Y->d_sliceEnt; t.nL = Y->d_nL; t.nU = Y->d_nU;
                    t.col = Y->d_col; t.sliceW = Y->d_sliceW; t.sliceT = Y->coop ? Y->d_sliceT : nullptr; t.rowIdx = Y->d_rowIdx;
                    LY.val[j - 1] = lv[j];
                }
            }
        }
    }
    if (!lay && gs_tasks_ensure(a, k)) return -1;
    auto it = a->gsTasks.find(lay ? k + 16 : k);
    if (it->second.n < 0) return 1;   // not pipelinable
    SliceTab T{a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_nL, a->d_nU, a->d_col};
    if (a->nCoopSlices) T.sliceT = a->d_sliceT;
    if (ctx->p2pGate)
    {
        T.gate = a->d_gateF;
        T.sliceDone = a->d_sliceDone;
    }
    const int nTasks = it->second.n;
    const int nChunks = cdiv(nTasks, P2P_CHUNK);
    // k sweeps are in flight at once: keep the same look-ahead (in levels) as a single sweep
    int bpc = ctx->p2pBlocksPerCU * k;
    if (bpc > ctx->p2pMaxBlocksPerCU) bpc = ctx->p2pMaxBlocksPerCU;
    if (gs_pu_slots(a) > 8 && bpc > 3) bpc = 3;   // 48 KB of LDS per workgroup
    // The kernel's 168 VGPRs hold three workgroups per CU, and every resident wave beyond the front is one more poller on the
    // memory system.  Levels that are bound by the hand-off (few slices per dependency level) run faster with TWO per CU, wide
    // ones - bound by how many tasks are in flight - with three: 12.7 M-cell motorBike mesh, bandCompression: 12.7 M cells x 2
    // sweeps (485 slices per level) 2.81 / 2.62 ms with 2 / 3 per CU, 6.3 M x 2 2.64 / 2.67, 3.1 M x 3 (75 per level) 3.55 / 4.29
    // (one per CU: 3.75 / 3.28 / 3.42); snappyHexMesh's numbering 2.28 / 1.95, 1.62 / 1.66, 2.62 / 3.34
    // (profiles/r05_gsm_bpc_probe.log).  A build whose kernel happened to need 169 VGPRs (two per CU) hid this until round 5.
    if (!ctx->p2pBpcForced)
        bpc = std::min(bpc, (double)a->nSlices / std::max(1, a->nLevels) * k >= ctx->gsmWideSlices ? 3 : 2);
    if (getenv("LDU_GSM_BPC")) bpc = std::max(1, atoi(getenv("LDU_GSM_BPC")));
    int grid = ctx->numCUs * bpc * 256 / P2P_BLK;
    if (grid > nChunks) grid = nChunks;
    if (grid < 1) grid = 1;
    if (P.gen != ctx->p2pGen)
    {
        LDU_CHECK_HIP(hipMemsetAsync(P.d_ticket, 0, sizeof(unsigned) * 64, s));
        P.ticketBase = 0;
        P.doneBase = 0;
        P.gen = ctx->p2pGen;
    }
    // tags tag0 .. tag0+k-1; keep them away from 0 and from wrapping inside one launch
    if (P.epoch > 0xffffff00u)
    {
        LDU_CHECK_HIP(hipMemsetAsync(P.d_granule, 0, sizeof(uint4) * (size_t)(a->nCells + 1), s));
        if (P.d_X) LDU_CHECK_HIP(hipMemsetAsync(P.d_X, 0, sizeof(uint4) * (size_t)(a->nCells + 1), s));
        P.epoch = 0;
    }
    const unsigned tag0 = P.epoch + 1;
    P.epoch += (unsigned)k;
    ctx->profStart(a, 4);   // "gs_multi": one launch = k pipelined sweeps
    T.sliceW = a->d_sliceW;
    if (!lay && it->second.d_slabTasks && use_slab(a, 2, k))
    {
        SliceTab TS{a->d_sliceRow, a->d_sliceCnt, a->d_sliceEnt, a->d_nL, a->d_nU, a->d_colX};
        if (a->nCoopSlices) TS.sliceT = a->d_sliceT;
        SlabCtl C;
        slab_ctl(a, P, C);
        C.list = it->second.d_slabTasks;
        for (int i = 0; i <= 8; i++) C.start[i] = it->second.slabStart[i];
        const int puSlots = gs_pu_slots(a);
        const size_t puBytes = sizeof(double) * (size_t)(P2P_BLK / LDU_WAVE) * puSlots * LDU_WAVE;
        int sbpc = slab_bpc(a, k);
        if (puSlots > 8) sbpc = std::min(sbpc, 3);   // 48 KB of LDS per workgroup
        const int sgrid = ctx->numCUs * sbpc;
        for (int i = 0; i < a->nSlabs; i++)
            C.window[i] = p2p_window(a, cdiv(it->second.slabStart[i + 1] - it->second.slabStart[i], P2P_CHUNK), k,
                                     sgrid / std::max(1, a->nSlabs), a->slabLevelSpan[i]);
        sweep_slab_gs_multi_kernel<<<sgrid, P2P_BLK, puBytes, s>>>(TS, C, k, P.d_granule, tag0,
            ctx->d_abort, psi, rhs, diag, val, puSlots);
        ctx->profStop(a, 4);
        LDU_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int window = p2p_window(a, nChunks, k, grid);
    const int puSlots = gs_pu_slots(a);
    // (held to 128 VGPRs = 4 waves per SIMD instead of 168 / 3 the kernel spills in its hot path: two sweeps of the 12.7 M-cell
    //  level 3.03 -> 5.67 ms, profiles/r05_gsm_wpe4_probe.log)
    if (lay)
        sweep_p2p_gs_multi_lay_kernel<<<grid, P2P_BLK, sizeof(double) * (size_t)(P2P_BLK / LDU_WAVE) * puSlots * LDU_WAVE, s>>>(
            T, it->second.d_tasks, nTasks, nChunks, k, P.d_ticket,
            P.ticketBase, window, P.doneBase, P.d_granule, tag0, ctx->d_abort, psi, rhs, diag, val, puSlots, LY);
    else
        sweep_p2p_gs_multi_kernel<<<grid, P2P_BLK, sizeof(double) * (size_t)(P2P_BLK / LDU_WAVE) * puSlots * LDU_WAVE, s>>>(
            T, it->second.d_tasks, nTasks, nChunks, k, P.d_ticket,
            P.ticketBase, window, P.doneBase, P.d_granule, tag0, ctx->d_abort, psi, rhs, diag, val, puSlots);
    ctx->profStop(a, 4);
    P.ticketBase += (unsigned)(nChunks + grid);
    if (window) P.doneBase += (unsigned)nChunks;
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

static int launch_sweep_p2p(ldu_addr* a, const SweepArgs& g, hipStream_t s)
{
    switch (g.mode)
    {
    case SW_TRI_FWD: return launch_p2p<SW_TRI_FWD, false>(a, g, s);
    case SW_TRI_BWD: return launch_p2p<SW_TRI_BWD, true>(a, g, s);
    case SW_RD:      return launch_p2p<SW_RD, false>(a, g, s);
    case SW_GS_FWD:  return launch_p2p<SW_GS_FWD, false>(a, g, s);
    case SW_GS_BWD:  return launch_p2p<SW_GS_BWD, true>(a, g, s);
    case SW_TRI_FWD_T: return launch_p2p<SW_TRI_FWD_T, false>(a, g, s);
    case SW_TRI_BWD_T: return launch_p2p<SW_TRI_BWD_T, true>(a, g, s);
    case SW_RD_T:      return launch_p2p<SW_RD_T, false>(a, g, s);
    case SW_GS_FWD_T:  return launch_p2p<SW_GS_FWD_T, false>(a, g, s);
    }
    return -1;
}

// Level-kernel engine: a sweep is hundreds of dependent launches: capture once per (mode,
// pointers) into a hipGraph and replay.  Kept as the fallback / cross-check of the
// point-to-point engine (LDU_SWEEP=levels).
int k_sweep(ldu_addr* a, const SweepArgs& g)
{
    ldu_ctx* ctx = a->ctx;
    hipStream_t s = g.stream ? g.stream : ctx->stream;
    if (a->nCells == 0) return 0;
    const int bm = sw_base(g.mode);
    const int cat = (bm == SW_GS_FWD || bm == SW_GS_BWD) ? LDU_PROF_GS_SWEEP
                    : (bm == SW_RD ? 7 : LDU_PROF_TRI_SWEEP);
    if (ctx->sweepP2P)
    {
        if (s == ctx->stream) ctx->profStart(a, cat);
        int rc = k_sweep_cluster(a, g, s);
        if (rc > 0) rc = launch_sweep_p2p(a, g, s);
        if (s == ctx->stream) ctx->profStop(a, cat);
        return rc;
    }
    if (!ctx->useGraphs || a->segs.size() <= 2)
    {
        if (s == ctx->stream) ctx->profStart(a, cat);
        int rc = launch_sweep(a, g, s);
        if (s == ctx->stream) ctx->profStop(a, cat);
        return rc;
    }

    char key[256];
    snprintf(key, sizeof(key), "%d|%p|%p|%p|%p|%p|%p", g.mode, (void*)g.w, (const void*)g.rhs,
             (const void*)g.scale, (const void*)g.val, (const void*)g.val2, (void*)g.aux);
    auto it = a->graphs.find(key);
    if (it == a->graphs.end())
    {
        hipGraph_t graph = nullptr;
        LDU_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = launch_sweep(a, g, s);
        hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc || e != hipSuccess)
        {
            ldu_set_error("sweep graph capture failed");
            return -1;
        }
        hipGraphExec_t exec = nullptr;
        LDU_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        if (a->graphs.size() > 64)
        {
            for (auto& kv : a->graphs) (void)hipGraphExecDestroy(kv.second);
            a->graphs.clear();
        }
        it = a->graphs.emplace(key, exec).first;
    }
    if (s == ctx->stream) ctx->profStart(a, cat);
    LDU_CHECK_HIP(hipGraphLaunch(it->second, s));
    if (s == ctx->stream) ctx->profStop(a, cat);
    return 0;
}

// ---------------------------------------------------------------- coupled patches

__global__ void pack_kernel(int n, const int* __restrict__ pfCell, const double* __restrict__ x,
                            double* __restrict__ send)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) send[i] = x[pfCell[i]];
}
int k_pack_patches(ldu_addr* a, const double* x, hipStream_t s)
{
    if (!a->nPatchFaces) return 0;
    pack_kernel<<<ewGrid(a->nPatchFaces), BLK, 0, s>>>(a->nPatchFaces, a->d_pfCell, x, a->d_sendAll);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// processorFvPatchScalarField.C:125-128: result[faceCells[i]] -= coeffs[i]*pnf[i], applied per
// boundary row in (patch, face) order; sign<0 = negated coefficients (residual / smoothers).
__global__ void apply_patches_kernel(int nBRows, const int* __restrict__ bRow,
                                     const int* __restrict__ bStart, const int* __restrict__ bFace,
                                     const double* __restrict__ coeffs, const double* __restrict__ recv,
                                     double sign, double* __restrict__ result)
{
    for (int j = blockIdx.x * BLK + threadIdx.x; j < nBRows; j += gridDim.x * BLK)
    {
        const int r = bRow[j];
        double acc = result[r];
        for (int t = bStart[j]; t < bStart[j + 1]; t++)
        {
            const int i = bFace[t];
            const double c = sign < 0 ? -coeffs[i] : coeffs[i];
            acc -= c * recv[i];
        }
        result[r] = acc;
    }
}
int k_apply_patches(ldu_addr* a, double* result, const double* coeffs, double sign, hipStream_t s)
{
    if (!a->nPatchFaces) return 0;
    if (comm_wait_halo(a, s)) return -1;   // updateMatrixInterfaces: the received values are needed from here on
    apply_patches_kernel<<<ewGrid(a->nBRows), BLK, 0, s>>>(a->nBRows, a->d_bRow, a->d_bStart, a->d_bFace,
                                                          coeffs, a->d_recvAll, sign, result);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// out[r] = in[r] - sum coeffs*pnf for the boundary rows only (same order as apply_patches_kernel)
__global__ void apply_patches_from_kernel(int nBRows, const int* __restrict__ bRow,
                                          const int* __restrict__ bStart, const int* __restrict__ bFace,
                                          const double* __restrict__ coeffs, const double* __restrict__ recv,
                                          double sign, const double* __restrict__ in, double* __restrict__ out)
{
    for (int j = blockIdx.x * BLK + threadIdx.x; j < nBRows; j += gridDim.x * BLK)
    {
        const int r = bRow[j];
        double acc = in[r];
        for (int t = bStart[j]; t < bStart[j + 1]; t++)
        {
            const int i = bFace[t];
            const double c = sign < 0 ? -coeffs[i] : coeffs[i];
            acc -= c * recv[i];
        }
        out[r] = acc;
    }
}
int k_apply_patches_from(ldu_addr* a, double* out, const double* in, const double* coeffs, double sign, hipStream_t s)
{
    if (!a->nPatchFaces) return 0;
    if (comm_wait_halo(a, s)) return -1;
    apply_patches_from_kernel<<<ewGrid(a->nBRows), BLK, 0, s>>>(a->nBRows, a->d_bRow, a->d_bStart, a->d_bFace,
                                                               coeffs, a->d_recvAll, sign, in, out);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// sumA: sumA[pa[face]] -= pCoeffs[face] (lduMatrixATmul.C:187-199)
__global__ void sumA_patches_kernel(int nBRows, const int* __restrict__ bRow,
                                    const int* __restrict__ bStart, const int* __restrict__ bFace,
                                    const double* __restrict__ bou, double* __restrict__ sumA)
{
    for (int j = blockIdx.x * BLK + threadIdx.x; j < nBRows; j += gridDim.x * BLK)
    {
        const int r = bRow[j];
        double acc = sumA[r];
        for (int t = bStart[j]; t < bStart[j + 1]; t++) acc -= bou[bFace[t]];
        sumA[r] = acc;
    }
}
int k_sumA_patches(ldu_addr* a, double* sumA, const double* bou, hipStream_t s)
{
    if (!a->nPatchFaces) return 0;
    sumA_patches_kernel<<<ewGrid(a->nBRows), BLK, 0, s>>>(a->nBRows, a->d_bRow, a->d_bStart, a->d_bFace,
                                                         bou, sumA);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- elementwise

__global__ void ew_kernel(int n, int op, double* __restrict__ y, const double* __restrict__ a,
                          const double* __restrict__ b)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
    {
        switch (op)
        {
        case EW_COPY: y[i] = a[i]; break;
        case EW_SUB: y[i] = a[i] - b[i]; break;
        case EW_ADD_INPLACE: y[i] += a[i]; break;
        case EW_MUL_INPLACE: y[i] *= a[i]; break;
        case EW_ZERO: y[i] = 0.0; break;
        case EW_DIV: y[i] = a[i] / b[i]; break;
        case EW_MUL: y[i] = a[i] * b[i]; break;
        case EW_SUB_INPLACE: y[i] -= a[i]; break;
        }
    }
}
int k_ew(int n, int op, double* y, const double* a, const double* b, hipStream_t s)
{
    if (n <= 0) return 0;
    ew_kernel<<<ewGrid(n), BLK, 0, s>>>(n, op, y, a, b);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// PCG.C:134-149: pA = wA (first) or pA = wA + beta*pA, beta = wArA/wArAold (device scalars)
__global__ void pcg_update_p_kernel(int n, double* __restrict__ pA, const double* __restrict__ wA,
                                    const double* __restrict__ S, int cur, int prev, int first)
{
    if (first)
    {
        for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) pA[i] = wA[i];
    }
    else
    {
        const double beta = S[cur] / S[prev];
        for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
            pA[i] = wA[i] + beta * pA[i];
    }
}
int k_pcg_update_p(int n, double* pA, const double* wA, const double* scalars, int cur, int prev,
                   int first, hipStream_t s)
{
    if (n <= 0) return 0;
    pcg_update_p_kernel<<<ewGrid(n), BLK, 0, s>>>(n, pA, wA, scalars, cur, prev, first);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// PBiCG.C:144-162
__global__ void pbicg_update_p_kernel(int n, double* __restrict__ pA, const double* __restrict__ wA,
                                      double* __restrict__ pT, const double* __restrict__ wT,
                                      const double* __restrict__ S, int cur, int prev, int first)
{
    if (first)
    {
        for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
        {
            pA[i] = wA[i];
            pT[i] = wT[i];
        }
    }
    else
    {
        const double beta = S[cur] / S[prev];
        for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
        {
            pA[i] = wA[i] + beta * pA[i];
            pT[i] = wT[i] + beta * pT[i];
        }
    }
}
int k_pbicg_update_p(int n, double* pA, const double* wA, double* pT, const double* wT,
                     const double* scalars, int cur, int prev, int first, hipStream_t s)
{
    if (n <= 0) return 0;
    pbicg_update_p_kernel<<<ewGrid(n), BLK, 0, s>>>(n, pA, wA, pT, wT, scalars, cur, prev, first);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- reductions

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void block_partial(double acc, double* partials)
{
    __shared__ double lds[WPB];
    acc = wave_sum(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t = lds[0];
        for (int i = 1; i < WPB; i++) t += lds[i];
        partials[blockIdx.x] = t;
    }
}

// RED_NORMFACTOR: sum |a - t| + |b - t|, t = c[i]*avg (lduMatrixSolver.C:187-192), avg from scalars
template <int OP>
__global__ void __launch_bounds__(BLK)
reduce_partial_kernel(int n, const double* __restrict__ a, const double* __restrict__ b,
                      const double* __restrict__ c, const double* __restrict__ S,
                      double* __restrict__ partials)
{
    double acc = 0.0;
    double avg = 0.0;
    if (OP == RED_NORMFACTOR) avg = S[S_SUMPSI] / S[S_COUNT];
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
    {
        if (OP == RED_DOT) acc += a[i] * b[i];
        else if (OP == RED_SUMMAG) acc += fabs(a[i]);
        else if (OP == RED_SUM) acc += a[i];
        else if (OP == RED_NORMFACTOR)
        {
            const double t = c[i] * avg;
            acc += fabs(a[i] - t) + fabs(b[i] - t);
        }
    }
    block_partial(acc, partials);
}

// two dot products at once: (a.b , c.b)  -> GAMG scale (GAMGSolverScale.C:54-58)
__global__ void __launch_bounds__(BLK)
reduce_dot2_kernel(int n, const double* __restrict__ a, const double* __restrict__ b,
                   const double* __restrict__ c, double* __restrict__ partials, int stride)
{
    double acc0 = 0.0, acc1 = 0.0;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
    {
        acc0 += a[i] * b[i];
        acc1 += c[i] * b[i];
    }
    __shared__ double lds[2 * WPB];
    acc0 = wave_sum(acc0);
    acc1 = wave_sum(acc1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { lds[wave] = acc0; lds[WPB + wave] = acc1; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t0 = lds[0], t1 = lds[WPB];
        for (int i = 1; i < WPB; i++) { t0 += lds[i]; t1 += lds[WPB + i]; }
        partials[blockIdx.x] = t0;
        partials[stride + blockIdx.x] = t1;
    }
}

__global__ void __launch_bounds__(BLK)
reduce_final_kernel(const double* __restrict__ partials, int nBlocks, double* __restrict__ S, int slot,
                    double addend, int nOut, int stride)
{
    __shared__ double lds[BLK];
    for (int o = 0; o < nOut; o++)
    {
        double acc = 0.0;
        for (int i = threadIdx.x; i < nBlocks; i += BLK) acc += partials[o * stride + i];
        lds[threadIdx.x] = acc;
        __syncthreads();
        for (int w = BLK / 2; w > 0; w >>= 1)
        {
            if (threadIdx.x < w) lds[threadIdx.x] += lds[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) S[slot + o] = lds[0] + addend;
        __syncthreads();
    }
}

static inline int redGrid(ldu_ctx* ctx, long n)
{
    int g = cdiv(n, BLK * 4);
    if (g < 1) g = 1;
    if (g > ctx->maxRedBlocks) g = ctx->maxRedBlocks;
    return g;
}

int k_reduce(ldu_ctx* ctx, int n, int op, const double* a, const double* b, const double* c,
             const double* d, int slot, hipStream_t s)
{
    (void)d;
    const int g = redGrid(ctx, n);
    const double* S = ctx->S();
    switch (op)
    {
    case RED_DOT: reduce_partial_kernel<RED_DOT><<<g, BLK, 0, s>>>(n, a, b, c, S, ctx->d_partials); break;
    case RED_SUMMAG: reduce_partial_kernel<RED_SUMMAG><<<g, BLK, 0, s>>>(n, a, b, c, S, ctx->d_partials); break;
    case RED_SUM: reduce_partial_kernel<RED_SUM><<<g, BLK, 0, s>>>(n, a, b, c, S, ctx->d_partials); break;
    case RED_NORMFACTOR:
        reduce_partial_kernel<RED_NORMFACTOR><<<g, BLK, 0, s>>>(n, a, b, c, S, ctx->d_partials); break;
    case RED_DOT2:
        reduce_dot2_kernel<<<g, BLK, 0, s>>>(n, a, b, c, ctx->d_partials, ctx->maxRedBlocks); break;
    default: return -1;
    }
    reduce_final_kernel<<<1, BLK, 0, s>>>(ctx->d_partials, g, ctx->S(), slot, 0.0,
                                         op == RED_DOT2 ? 2 : 1, ctx->maxRedBlocks);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// PCG.C:159-172 / PBiCG.C:170-188: singular test, psi += alpha pA, rA -= alpha wA (rT -= alpha wT),
// and the partial sums of |rA| for the residual - one pass.
__global__ void __launch_bounds__(BLK)
pcg_update_xr_kernel(int n, double* __restrict__ psi, double* __restrict__ rA,
                     const double* __restrict__ pA, const double* __restrict__ wA,
                     double* __restrict__ rT, const double* __restrict__ wT, double* __restrict__ S,
                     int cur, double* __restrict__ partials)
{
    const double wApA = S[S_WAPA];
    const bool singular = fabs(wApA) / S[S_NORM] < 1e-300;   // solverPerformance::vsmall_
    double acc = 0.0;
    // a speculatively queued iteration behind the last one of the solve: nothing is touched (k_krylov_decide)
    if (S[S_STOP] != 0.0) { block_partial(acc, partials); return; }
    if (singular)
    {
        if (blockIdx.x == 0 && threadIdx.x == 0) S[S_SINGULAR] = 1.0;
        for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) acc += fabs(rA[i]);
    }
    else
    {
        const double alpha = S[cur] / wApA;
        for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
        {
            psi[i] += alpha * pA[i];
            const double r = rA[i] - alpha * wA[i];
            rA[i] = r;
            if (rT) rT[i] -= alpha * wT[i];
            acc += fabs(r);
        }
    }
    block_partial(acc, partials);
}

int k_pcg_update_xr(ldu_ctx* ctx, int n, double* psi, double* rA, const double* pA, const double* wA,
                    double* rT, const double* wT, int cur, hipStream_t s)
{
    const int g = redGrid(ctx, n);
    pcg_update_xr_kernel<<<g, BLK, 0, s>>>(n, psi, rA, pA, wA, rT, wT, ctx->S(), cur,
                                           ctx->d_partials);
    reduce_final_kernel<<<1, BLK, 0, s>>>(ctx->d_partials, g, ctx->S(), S_RES, 0.0, 1,
                                         ctx->maxRedBlocks);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// The loop condition of PCG.C:174-181 / PBiCG.C:181-188 on the device, after the residual of iteration `it` was formed:
// finalResidual = sum|rA| / normFactor; converged = finalResidual < Tolerance || (RelTolerance > small &&
// finalResidual < RelTolerance*initialResidual) (lduMatrixSolver.C / SolverPerformance.C:37-70); the same IEEE operations the
// host performs on the same doubles, so both sides take the same decision.  A set flag stays set.
__global__ void krylov_decide_kernel(double* __restrict__ S, double tolerance, double relTol, int it, int maxIter)
{
    if (S[S_STOP] != 0.0) return;
    const double fin = S[S_RES] / S[S_NORM];
    const bool conv = fin < tolerance || (relTol > 1e-20 && fin < relTol * S[S_INIT]);
    S[S_STOP] = (S[S_SINGULAR] != 0.0 || conv || !(it < maxIter)) ? 1.0 : 0.0;
}
int k_krylov_decide(ldu_ctx* ctx, double tolerance, double relTol, int it, int maxIter, hipStream_t s)
{
    krylov_decide_kernel<<<1, 1, 0, s>>>(ctx->S(), tolerance, relTol, it, maxIter);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// GAMGSolverScale.C:62-74: sf = num/stabilise(den, VSMALL); field = sf*field + (source - sf*Acf)/D
__global__ void gamg_scale_update_kernel(int n, double* __restrict__ field,
                                         const double* __restrict__ source,
                                         const double* __restrict__ Acf, const double* __restrict__ diag,
                                         const double* __restrict__ S)
{
    const double den = S[S_SCALE_DEN];
    const double sf = S[S_SCALE_NUM] / (den >= 0 ? den + 1e-300 : den - 1e-300);
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
        field[i] = sf * field[i] + (source[i] - sf * Acf[i]) / diag[i];
}
int k_gamg_scale_update(int n, double* field, const double* source, const double* Acf,
                        const double* diag, const double* scalars, hipStream_t s)
{
    if (n <= 0) return 0;
    gamg_scale_update_kernel<<<ewGrid(n), BLK, 0, s>>>(n, field, source, Acf, diag, scalars);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// GAMGSolverInterpolate.C:78-82: psi = -Apsi/diag
__global__ void neg_div_kernel(int n, double* __restrict__ psi, const double* __restrict__ Apsi,
                               const double* __restrict__ diag)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) psi[i] = -Apsi[i] / diag[i];
}
int k_neg_div(int n, double* psi, const double* Apsi, const double* diag, hipStream_t s)
{
    if (n <= 0) return 0;
    neg_div_kernel<<<ewGrid(n), BLK, 0, s>>>(n, psi, Apsi, diag);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- GAMG transfer

// GAMGAgglomerationTemplates.C:31-59: cf = 0; cf[map[i]] += ff[i] in ascending fine index ->
// per coarse cell: children (ascending ORIGINAL fine index) summed left to right.
__global__ void restrict_kernel(int nCoarse, const int* __restrict__ childStart,
                                const int* __restrict__ child, const double* __restrict__ fine,
                                double* __restrict__ coarse)
{
    for (int c = blockIdx.x * BLK + threadIdx.x; c < nCoarse; c += gridDim.x * BLK)
    {
        double acc = 0.0;
        for (int t = childStart[c]; t < childStart[c + 1]; t++) acc += fine[child[t]];
        coarse[c] = acc;
    }
}
int k_restrict(int nCoarse, const int* childStart, const int* child, const double* fine, double* coarse,
               hipStream_t s)
{
    if (nCoarse <= 0) return 0;
    restrict_kernel<<<ewGrid(nCoarse), BLK, 0, s>>>(nCoarse, childStart, child, fine, coarse);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// GAMGAgglomerationTemplates.C:87-100: ff[i] = cf[map[i]]
__global__ void prolong_kernel(int nFine, const int* __restrict__ map, const double* __restrict__ coarse,
                               double* __restrict__ fine)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < nFine; i += gridDim.x * BLK) fine[i] = coarse[map[i]];
}
int k_prolong(int nFine, const int* map, const double* coarse, double* fine, hipStream_t s)
{
    if (nFine <= 0) return 0;
    prolong_kernel<<<ewGrid(nFine), BLK, 0, s>>>(nFine, map, coarse, fine);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// GAMGSolverAgglomerateMatrix.C:148-205 as gathers in ascending fine-face order (all arrays in
// ORIGINAL numbering of their level).
__global__ void agg_faces_kernel(int nCoarseFaces, const int* __restrict__ cfStart,
                                 const int* __restrict__ cfFine, const unsigned char* __restrict__ cfFlip,
                                 const double* __restrict__ fineUpper, const double* __restrict__ fineLower,
                                 double* __restrict__ coarseUpper, double* __restrict__ coarseLower,
                                 bool sym)
{
    for (int c = blockIdx.x * BLK + threadIdx.x; c < nCoarseFaces; c += gridDim.x * BLK)
    {
        double up = 0.0, lo = 0.0;
        for (int t = cfStart[c]; t < cfStart[c + 1]; t++)
        {
            const int f = cfFine[t];
            if (sym) up += fineUpper[f];
            else if (!cfFlip[t]) { up += fineUpper[f]; lo += fineLower[f]; }
            else { up += fineLower[f]; lo += fineUpper[f]; }
        }
        coarseUpper[c] = up;
        if (!sym) coarseLower[c] = lo;
    }
}
__global__ void agg_diag_kernel(int nCoarseCells, const int* __restrict__ childStartO,
                                const int* __restrict__ childO, const int* __restrict__ ccStart,
                                const int* __restrict__ ccFine, const double* __restrict__ fineDiag,
                                const double* __restrict__ fineUpper, const double* __restrict__ fineLower,
                                double* __restrict__ coarseDiag, bool sym)
{
    for (int c = blockIdx.x * BLK + threadIdx.x; c < nCoarseCells; c += gridDim.x * BLK)
    {
        double acc = 0.0;
        for (int t = childStartO[c]; t < childStartO[c + 1]; t++) acc += fineDiag[childO[t]];
        for (int t = ccStart[c]; t < ccStart[c + 1]; t++)
        {
            const int f = ccFine[t];
            if (sym) acc += 2 * fineUpper[f];
            else acc += fineUpper[f] + fineLower[f];
        }
        coarseDiag[c] = acc;
    }
}
int k_agglomerate_coeffs(int nCoarseFaces, const int* cfStart, const int* cfFine, const unsigned char* cfFlip,
                         int nCoarseCells, const int* ccStart, const int* ccFine,
                         const int* childStartO, const int* childO,
                         const double* fineDiag, const double* fineUpper, const double* fineLower,
                         double* coarseDiag, double* coarseUpper, double* coarseLower, bool sym,
                         hipStream_t s)
{
    if (nCoarseFaces > 0)
        agg_faces_kernel<<<ewGrid(nCoarseFaces), BLK, 0, s>>>(nCoarseFaces, cfStart, cfFine, cfFlip,
            fineUpper, fineLower, coarseUpper, coarseLower, sym);
    if (nCoarseCells > 0)
        agg_diag_kernel<<<ewGrid(nCoarseCells), BLK, 0, s>>>(nCoarseCells, childStartO, childO, ccStart,
            ccFine, fineDiag, fineUpper, fineLower, coarseDiag, sym);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// GAMGInterface::agglomerateCoeffs (GAMGInterface.C:61-75): coarse[fra[ffi]] += fine[ffi] in
// ascending ffi, as a gather per coarse patch face
__global__ void patch_agg_kernel(int nCoarse, const int* __restrict__ start, const int* __restrict__ fine,
                                 const double* __restrict__ fBou, const double* __restrict__ fInt,
                                 double* __restrict__ cBou, double* __restrict__ cInt)
{
    for (int c = blockIdx.x * BLK + threadIdx.x; c < nCoarse; c += gridDim.x * BLK)
    {
        double b = 0.0, i = 0.0;
        for (int t = start[c]; t < start[c + 1]; t++)
        {
            b += fBou[fine[t]];
            i += fInt[fine[t]];
        }
        cBou[c] = b;
        cInt[c] = i;
    }
}
int k_patch_agglomerate(int nCoarse, const int* start, const int* fine, const double* fBou,
                        const double* fInt, double* cBou, double* cInt, hipStream_t s)
{
    if (nCoarse <= 0) return 0;
    patch_agg_kernel<<<ewGrid(nCoarse), BLK, 0, s>>>(nCoarse, start, fine, fBou, fInt, cBou, cInt);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- fv stencils (original numbering)

// surfaceInterpolationScheme.C:293-296: sf = lambda*(vf[P]-vf[N]) + vf[N]
// One lane per (face, component): the stores of a wave are one contiguous run, the 24 / 72 bytes of a cell's value arrive
// through adjacent lanes (a lane per face walking its components touched every cache line nComp times: 37 % of HBM peak
// for vectors against 61 % for scalars).
template <int NC>
__global__ void fv_interpolate_kernel(long n, const int* __restrict__ P, const int* __restrict__ N,
                                      const double* __restrict__ lambda, const double* __restrict__ vf,
                                      double* __restrict__ sf)
{
    // four elements per lane and trip: the two dependent loads (face -> cells -> values) of all four are in flight together
    const long stride = (long)gridDim.x * BLK;
    for (long e0 = (long)blockIdx.x * BLK + threadIdx.x; e0 < n; e0 += 4 * stride)
    {
        long p[4], q[4];
        double lam[4], a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const long e = e0 + k * stride < n ? e0 + k * stride : e0;
            const long f = e / NC;
            const int c = (int)(e - f * NC);
            p[k] = (long)P[f] * NC + c; q[k] = (long)N[f] * NC + c;
            lam[k] = lambda[f];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = vf[p[k]]; b[k] = vf[q[k]]; }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (e0 + k * stride < n) sf[e0 + k * stride] = lam[k] * (a[k] - b[k]) + b[k];
    }
}
__global__ void fv_interpolate_any_kernel(long n, int nComp, const int* __restrict__ P, const int* __restrict__ N,
                                          const double* __restrict__ lambda, const double* __restrict__ vf,
                                          double* __restrict__ sf)
{
    for (long e = (long)blockIdx.x * BLK + threadIdx.x; e < n; e += (long)gridDim.x * BLK)
    {
        const long f = e / nComp;
        const int c = (int)(e - f * nComp);
        const long p = P[f], q = N[f];
        const double lam = lambda[f];
        const double a = vf[p * nComp + c], b = vf[q * nComp + c];
        sf[e] = lam * (a - b) + b;
    }
}
int k_fv_interpolate(ldu_addr* a, int nComp, const double* lambdas, const double* vf, double* sf,
                     hipStream_t s)
{
    if (a->nFaces == 0) return 0;
    const long n = (long)a->nFaces * nComp;
    int grid = (int)std::min<long>((n + BLK - 1) / BLK, 1 << 20);
    if (nComp == 1 || nComp == 3 || nComp == 6 || nComp == 9) grid = std::max(1, (grid + 3) / 4);   // (four elements per lane)
    switch (nComp)
    {
    case 1: fv_interpolate_kernel<1><<<grid, BLK, 0, s>>>(n, a->d_l, a->d_u, lambdas, vf, sf); break;
    case 3: fv_interpolate_kernel<3><<<grid, BLK, 0, s>>>(n, a->d_l, a->d_u, lambdas, vf, sf); break;
    case 6: fv_interpolate_kernel<6><<<grid, BLK, 0, s>>>(n, a->d_l, a->d_u, lambdas, vf, sf); break;
    case 9: fv_interpolate_kernel<9><<<grid, BLK, 0, s>>>(n, a->d_l, a->d_u, lambdas, vf, sf); break;
    default: fv_interpolate_any_kernel<<<grid, BLK, 0, s>>>(n, nComp, a->d_l, a->d_u, lambdas, vf, sf); break;
    }
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// fvcSurfaceIntegrate.C:56-60,75 / gaussGrad.C:82-88,105 as a cell gather in the reference's
// face order: a cell receives -= from faces where it is the neighbour and += from owned faces,
// interleaved by ascending face index == (neighbour faces ascending) then (owned faces ascending).
// sfVec != NULL: contribution = Sf[f]*ssf[f] (gaussGrad, nComp = 3, ssf scalar per face)
__global__ void fv_surfaceIntegrate_kernel(int nCells, int nComp, const int* __restrict__ losortStart,
                                           const int* __restrict__ losort,
                                           const int* __restrict__ ownerStart,
                                           const double* __restrict__ ssf,
                                           const double* __restrict__ sfVec,
                                           const double* __restrict__ V, double* __restrict__ out)
{
    for (int c = blockIdx.x * BLK + threadIdx.x; c < nCells; c += gridDim.x * BLK)
    {
        for (int k = 0; k < nComp; k++)
        {
            double acc = 0.0;
            for (int t = losortStart[c]; t < losortStart[c + 1]; t++)
            {
                const int f = losort[t];
                acc -= sfVec ? sfVec[(long)f * 3 + k] * ssf[f] : ssf[(long)f * nComp + k];
            }
            for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
                acc += sfVec ? sfVec[(long)f * 3 + k] * ssf[f] : ssf[(long)f * nComp + k];
            out[(long)c * nComp + k] = acc / V[c];
        }
    }
}
int k_fv_surfaceIntegrate(ldu_addr* a, int nComp, const double* ssf, const double* sfVec, const double* V,
                          double* out, hipStream_t s)
{
    if (a->nCells == 0) return 0;
    fv_surfaceIntegrate_kernel<<<ewGrid(a->nCells), BLK, 0, s>>>(a->nCells, nComp, a->d_losortStart,
        a->d_losort, a->d_ownerStart, ssf, sfVec, V, out);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// snGradScheme.C:139-143
__global__ void fv_snGrad_kernel(int nFaces, const int* __restrict__ own, const int* __restrict__ nei,
                                 const double* __restrict__ delta, const double* __restrict__ vf,
                                 double* __restrict__ ssf)
{
    for (int f = blockIdx.x * BLK + threadIdx.x; f < nFaces; f += gridDim.x * BLK)
        ssf[f] = delta[f] * (vf[nei[f]] - vf[own[f]]);
}
int k_fv_snGrad(ldu_addr* a, const double* delta, const double* vf, double* ssf, hipStream_t s)
{
    if (a->nFaces == 0) return 0;
    fv_snGrad_kernel<<<ewGrid(a->nFaces), BLK, 0, s>>>(a->nFaces, a->d_l, a->d_u, delta, vf, ssf);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// lduMatrix::negSumDiag (lduMatrixOperations.C:50-64): diag[l] -= lower; diag[u] -= upper in face
// order, from a zero diagonal: row gather (neighbour faces asc. then owned faces asc.)
__global__ void fv_negSumDiag_kernel(int nCells, const int* __restrict__ losortStart,
                                     const int* __restrict__ losort, const int* __restrict__ ownerStart,
                                     const double* __restrict__ lower, const double* __restrict__ upper,
                                     double* __restrict__ diag)
{
    for (int c = blockIdx.x * BLK + threadIdx.x; c < nCells; c += gridDim.x * BLK)
    {
        double acc = 0.0;
        // (four faces at a time: the index loads, then the coefficient loads, are in flight together)
        const int t1 = losortStart[c + 1];
        for (int t = losortStart[c]; t < t1; t += 4)
        {
            int f[4];
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) f[i] = losort[t + i < t1 ? t + i : t1 - 1];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = upper[f[i]];
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (t + i < t1) acc -= v[i];
        }
        const int f1 = ownerStart[c + 1];
        for (int f = ownerStart[c]; f < f1; f += 4)
        {
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = lower[f + i < f1 ? f + i : f1 - 1];
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (f + i < f1) acc -= v[i];
        }
        diag[c] = acc;
    }
}
int k_fv_negSumDiag(ldu_addr* a, const double* lower, const double* upper, double* diag, hipStream_t s)
{
    if (a->nCells == 0) return 0;
    fv_negSumDiag_kernel<<<ewGrid(a->nCells), BLK, 0, s>>>(a->nCells, a->d_losortStart, a->d_losort,
        a->d_ownerStart, lower, upper, diag);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// gaussLaplacianScheme.C:63: upper = deltaCoeffs*gammaMagSf
__global__ void fv_lap_kernel(int n, const double* __restrict__ delta, const double* __restrict__ g,
                              double* __restrict__ upper)
{
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n; f += gridDim.x * BLK) upper[f] = delta[f] * g[f];
}
int k_fv_laplacian_coeffs(int nFaces, const double* delta, const double* gammaMagSf, double* upper,
                          hipStream_t s)
{
    if (nFaces == 0) return 0;
    fv_lap_kernel<<<ewGrid(nFaces), BLK, 0, s>>>(nFaces, delta, gammaMagSf, upper);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// gaussConvectionScheme.C:87-88: lower = -w*phi; upper = lower + phi
__global__ void fv_div_kernel(int n, const double* __restrict__ w, const double* __restrict__ phi,
                              double* __restrict__ lower, double* __restrict__ upper)
{
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n; f += gridDim.x * BLK)
    {
        const double lo = -w[f] * phi[f];
        lower[f] = lo;
        upper[f] = lo + phi[f];
    }
}
int k_fv_div_coeffs(int nFaces, const double* w, const double* phi, double* lower, double* upper,
                    hipStream_t s)
{
    if (nFaces == 0) return 0;
    fv_div_kernel<<<ewGrid(nFaces), BLK, 0, s>>>(nFaces, w, phi, lower, upper);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

// gaussLaplacianScheme.C:63 / gaussConvectionScheme.C:87-88 AND lduMatrix::negSumDiag in one pass: a tile of BLK consecutive
// cells owns one contiguous face range (faces are ordered by owner); its two input arrays are read once by coalesced
// loads, the coefficients are written from there and kept in LDS, and every cell sums its owned faces - and the
// neighbour-side faces the tile owns too - from LDS.  Neighbour-side faces of other tiles: the same coefficient formed
// again from its two inputs (same arithmetic, no contraction: bit-identical to what the owning tile wrote).  Same order
// of subtractions per cell as fv_negSumDiag_kernel.  MODE 0: upper = delta*gamma (symmetric); MODE 1: lower = -w*phi,
// upper = lower + phi.
#define FVC_MAXF 1024
template <int MODE>
__global__ void __launch_bounds__(BLK)
fv_coeffs_diag_tile_kernel(int nCells, const int* __restrict__ losortStart, const int* __restrict__ losort,
                           const int* __restrict__ ownerStart, const double* __restrict__ A, const double* __restrict__ B,
                           double* __restrict__ lower, double* __restrict__ upper, double* __restrict__ diag)
{
    __shared__ double sUp[FVC_MAXF];
    __shared__ double sLo[MODE == 1 ? FVC_MAXF : 1];
    const int nTiles = (nCells + BLK - 1) / BLK;
    const int per = (nTiles + 7) >> 3;                      // XCD x walks the x-th eighth of the tiles in order
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= nTiles) return;
    const int c0 = tile * BLK;
    const int cEnd = c0 + BLK < nCells ? c0 + BLK : nCells;
    const int fA = ownerStart[c0];
    const int nOwnAll = ownerStart[cEnd] - fA;
    const int nOwn = nOwnAll < FVC_MAXF ? nOwnAll : FVC_MAXF;
    for (int e = threadIdx.x; e < nOwnAll; e += BLK)
    {
        const size_t f = (size_t)fA + e;
        double lo, up;
        if (MODE == 0) { up = A[f] * B[f]; lo = up; }
        else { const double ph = B[f]; lo = -A[f] * ph; up = lo + ph; lower[f] = lo; }
        upper[f] = up;
        if (e < nOwn) { sUp[e] = up; if (MODE == 1) sLo[e] = lo; }
    }
    __syncthreads();
    const int c = c0 + threadIdx.x;
    if (c >= nCells) return;
    double acc = 0.0;
    const int t1 = losortStart[c + 1];
    for (int t = losortStart[c]; t < t1; t += 4)
    {
        int f[4];
        double v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) f[i] = losort[t + i < t1 ? t + i : t1 - 1];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const unsigned fl = (unsigned)(f[i] - fA);
            if (fl < (unsigned)nOwn) v[i] = sUp[fl];
            else if (MODE == 0) v[i] = A[f[i]] * B[f[i]];
            else { const double ph = B[f[i]]; const double lo = -A[f[i]] * ph; v[i] = lo + ph; }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (t + i < t1) acc -= v[i];
    }
    const int f1 = ownerStart[c + 1];
    for (int f = ownerStart[c]; f < f1; f++)
    {
        const unsigned fl = (unsigned)(f - fA);
        double v;
        if (fl < (unsigned)nOwn) v = MODE == 1 ? sLo[fl] : sUp[fl];
        else if (MODE == 0) v = A[f] * B[f];
        else v = -A[f] * B[f];
        acc -= v;
    }
    diag[c] = acc;
}

// 0 = done; 1 = not applicable (in-place arguments)
int k_fv_coeffs_diag(ldu_addr* a, int mode, const double* A, const double* B, double* lower, double* upper, double* diag,
                     hipStream_t s)
{
    if (a->nCells == 0) return 0;
    if (A == upper || B == upper || A == lower || B == lower) return 1;   // (the tiles re-read inputs other tiles write beside)
    const int nTiles = (a->nCells + BLK - 1) / BLK;
    const int grid = 8 * ((nTiles + 7) / 8);
    if (mode == 0)
        fv_coeffs_diag_tile_kernel<0><<<grid, BLK, 0, s>>>(a->nCells, a->d_losortStart, a->d_losort, a->d_ownerStart, A, B,
                                                           nullptr, upper, diag);
    else
        fv_coeffs_diag_tile_kernel<1><<<grid, BLK, 0, s>>>(a->nCells, a->d_losortStart, a->d_losort, a->d_ownerStart, A, B,
                                                           lower, upper, diag);
    LDU_CHECK_HIP(hipGetLastError());
    return 0;
}

int k_set_watchdog(unsigned long long budgetTicks, unsigned long long stallTicks)
{
    const unsigned long long v[2] = {budgetTicks, stallTicks};
    LDU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wait_budget), v, sizeof(v)));
    return 0;
}
