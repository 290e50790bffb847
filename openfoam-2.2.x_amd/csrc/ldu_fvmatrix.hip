// fvMatrix glue executed around every solve, scalar matrices:
// addBoundaryDiag / addBoundarySource (fvMatrix.C:116-178), setReference (:509-521), relax (:525-655),
// A (:722-746), H (fvScalarMatrix.C:209-237), flux (:865-943).  Everything stays in HBM between assembly and solve.
//
// Layout: ORIGINAL cell / face numbering (these run before the matrix is handed to the solver).
// One thread per cell; a cell walks its boundary faces in (patch, face) order - the order in which
// the reference's patch loops touch it - and its internal faces in the reference's face order
// (faces whose upper cell it is, ascending, then the faces it owns, ascending), so every result is
// bit-identical to the sequential loops (-ffp-contract=off).
#include <algorithm>
#include <vector>

#include "ldu_internal.hpp"

static bool dev_ptr(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice;
}

// host/device staging (same contract as the other fv entry points: any pointer may be host or device)
struct GlueBuf {
    hipStream_t s;
    std::vector<void*> owned;
    explicit GlueBuf(hipStream_t ss) : s(ss) {}
    ~GlueBuf() { for (void* p : owned) (void)hipFree(p); }
    const double* in(const double* user, size_t n)
    {
        if (!user || dev_ptr(user)) return user;
        double* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(double) * (n ? n : 1)) != hipSuccess) return nullptr;
        owned.push_back(d);
        (void)hipMemcpyAsync(d, user, sizeof(double) * n, hipMemcpyHostToDevice, s);
        return d;
    }
    double* inout(double* user, size_t n, bool copyIn)
    {
        if (dev_ptr(user)) return user;
        double* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(double) * (n ? n : 1)) != hipSuccess) return nullptr;
        owned.push_back(d);
        if (copyIn) (void)hipMemcpyAsync(d, user, sizeof(double) * n, hipMemcpyHostToDevice, s);
        return d;
    }
    int finish(double* user, double* dev, size_t n)
    {
        if (user != dev) LDU_CHECK_HIP(hipMemcpyAsync(user, dev, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        LDU_CHECK_HIP(hipStreamSynchronize(s));
        return 0;
    }
};

#define GLUE_BLK 256
static inline int glue_grid(int n) { return (n + GLUE_BLK - 1) / GLUE_BLK; }

// ---------------------------------------------------------------- kernels

// fvMatrix.C:116-131 (and addCmptAvBoundaryDiag for a scalar): diag[faceCells] += internalCoeffs
__global__ void __launch_bounds__(GLUE_BLK)
glue_addBoundaryDiag_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                            const double* __restrict__ iC, double* __restrict__ diag)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    const int b = cs[c], e = cs[c + 1];
    if (b == e) return;
    double d = diag[c];
    for (int j = b; j < e; j++) d += iC[cf[j]];
    diag[c] = d;
}

// fvMatrix.C:150-178
__global__ void __launch_bounds__(GLUE_BLK)
glue_addBoundarySource_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                              const unsigned char* __restrict__ coupled, const double* __restrict__ bC,
                              const double* __restrict__ pnf, int couples, double* __restrict__ source)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    const int b = cs[c], e = cs[c + 1];
    if (b == e) return;
    double s = source[c];
    for (int j = b; j < e; j++)
    {
        const int f = cf[j];
        if (!coupled[f]) s += bC[f];
        else if (couples) s += bC[f] * pnf[f];
    }
    source[c] = s;
}

// fvMatrix.C:525-655, Type = scalar
__global__ void __launch_bounds__(GLUE_BLK)
glue_relax_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                  const unsigned char* __restrict__ coupled, const double* __restrict__ iC,
                  const double* __restrict__ bC, const int* __restrict__ losortStart,
                  const int* __restrict__ losort, const int* __restrict__ ownerStart,
                  const double* __restrict__ upper, const double* __restrict__ lower, double alpha,
                  const double* __restrict__ psi, double* __restrict__ diag, double* __restrict__ source)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    const double D0 = diag[c];
    double D = D0;
    // sumMagOffDiag (lduMatrixOperations.C:67-83), this cell's updates in face order
    double sumOff = 0.0;
    for (int j = losortStart[c]; j < losortStart[c + 1]; j++) sumOff += fabs(lower[losort[j]]);
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++) sumOff += fabs(upper[f]);
    const int b = cs[c], e = cs[c + 1];
    for (int j = b; j < e; j++)
    {
        const int f = cf[j];
        if (coupled[f])
        {
            D += iC[f];
            sumOff += fabs(bC[f]);
        }
        else
            D += fabs(iC[f]);             // cmptMax(cmptMag(.)) of a scalar
    }
    D = fmax(fabs(D), sumOff);            // max(mag(D), sumOff)
    D /= alpha;
    for (int j = b; j < e; j++) D -= iC[cf[j]];   // component 0 / cmptMin of a scalar
    diag[c] = D;
    source[c] += (D - D0) * psi[c];
}

// fvMatrix.C:722-746: A = D()/V, D() = diag + cmptAv(internalCoeffs)
__global__ void __launch_bounds__(GLUE_BLK)
glue_A_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf, const double* __restrict__ iC,
              const double* __restrict__ diag, const double* __restrict__ V, double* __restrict__ A)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    double d = diag[c];
    for (int j = cs[c]; j < cs[c + 1]; j++) d += iC[cf[j]];
    A[c] = d / V[c];
}

// fvMatrix<scalar>::H = the scalar specialisation fvScalarMatrix.C:209-237 (no boundary-diagonal term);
// lduMatrix::H = lduMatrixTemplates.C:32-69: H = (source - sum_nbr lower*psi[l] - sum_own upper*psi[u] + boundary source)/V,
// per cell the neighbour-side faces (ascending face index) before the owned ones, four faces at a time (index ->
// coefficient, column -> psi, each stage's loads in flight together).
// The owned faces of a tile of GLUE_BLK consecutive cells are staged through LDS (one contiguous face range:
// upper, lower, neighbour label by coalesced loads): a face's coefficient and column come from LDS, only psi is gathered -
// two dependent global round trips per face instead of three (216^3: 0.324 -> 0.305 ms).  Neighbour-side faces owned by
// other tiles read their coefficient from global memory.  XCD-aware tile order (see xcd_tile).
#define GH_MAXF 1024
__global__ void __launch_bounds__(GLUE_BLK)
glue_H_tile_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                   const unsigned char* __restrict__ coupled, const double* __restrict__ bC, const double* __restrict__ pnf,
                   const int* __restrict__ losortStart, const int* __restrict__ losort,
                   const int* __restrict__ ownerStart, const int* __restrict__ l, const int* __restrict__ u,
                   const double* __restrict__ upper, const double* __restrict__ lower,
                   const double* __restrict__ psi, const double* __restrict__ source, const double* __restrict__ V,
                   double* __restrict__ H)
{
    __shared__ double sUp[GH_MAXF];
    __shared__ double sLo[GH_MAXF];
    __shared__ int sU[GH_MAXF];
    const int nTiles = (nCells + GLUE_BLK - 1) / GLUE_BLK;
    const int per = (nTiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= nTiles) return;
    const int c0 = tile * GLUE_BLK;
    const int cEnd = c0 + GLUE_BLK < nCells ? c0 + GLUE_BLK : nCells;
    const int fA = ownerStart[c0];
    int nOwn = ownerStart[cEnd] - fA;
    if (nOwn > GH_MAXF) nOwn = GH_MAXF;
    for (int e = threadIdx.x; e < nOwn; e += GLUE_BLK)
    {
        sUp[e] = upper[(size_t)fA + e];
        sLo[e] = lower[(size_t)fA + e];
        sU[e] = u[(size_t)fA + e];
    }
    __syncthreads();
    const int c = c0 + threadIdx.x;
    if (c >= nCells) return;
    double hl = 0.0;
    const int t1 = losortStart[c + 1];
    for (int t = losortStart[c]; t < t1; t += 4)
    {
        int f[4], col[4];
        double co[4], ps[4];
#pragma unroll
        for (int i = 0; i < 4; i++) f[i] = losort[t + i < t1 ? t + i : t1 - 1];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const unsigned fl = (unsigned)(f[i] - fA);
            co[i] = fl < (unsigned)nOwn ? sLo[fl] : lower[f[i]];
            col[i] = l[f[i]];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) ps[i] = psi[col[i]];
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (t + i < t1) hl -= co[i] * ps[i];
    }
    const int f1 = ownerStart[c + 1];
    for (int f = ownerStart[c]; f < f1; f += 4)
    {
        int col[4];
        double co[4], ps[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int g = f + i < f1 ? f + i : f1 - 1;
            const unsigned fl = (unsigned)(g - fA);
            if (fl < (unsigned)nOwn) { co[i] = sUp[fl]; col[i] = sU[fl]; }
            else { co[i] = upper[g]; col[i] = u[g]; }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) ps[i] = psi[col[i]];
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (f + i < f1) hl -= co[i] * ps[i];
    }
    double h = hl + source[c];
    for (int j = cs[c]; j < cs[c + 1]; j++)
    {
        const int f = cf[j];
        h += coupled[f] ? bC[f] * pnf[f] : bC[f];
    }
    H[c] = h / V[c];
}

// fvMatrix.C:865-943: internal faces = lduMatrix::faceH (lduMatrixTemplates.C:104-135)
__global__ void __launch_bounds__(GLUE_BLK)
glue_flux_internal_kernel(int nFaces, const int* __restrict__ l, const int* __restrict__ u,
                          const double* __restrict__ upper, const double* __restrict__ lower,
                          const double* __restrict__ psi, double* __restrict__ flux)
{
    const int f = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (f >= nFaces) return;
    flux[f] = upper[f] * psi[u[f]] - lower[f] * psi[l[f]];
}

// boundary faces: internalCoeffs*patchInternalField - (coupled ? boundaryCoeffs*patchNeighbourField : boundaryCoeffs)
__global__ void __launch_bounds__(GLUE_BLK)
glue_flux_boundary_kernel(int n, const int* __restrict__ faceCells, const unsigned char* __restrict__ coupled,
                          const double* __restrict__ iC, const double* __restrict__ bC,
                          const double* __restrict__ pnf, const double* __restrict__ psi,
                          double* __restrict__ flux)
{
    const int f = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (f >= n) return;
    const double ic = iC[f] * psi[faceCells[f]];
    const double nc = coupled[f] ? bC[f] * pnf[f] : bC[f];
    flux[f] = ic - nc;
}

__global__ void glue_setReference_kernel(int celli, double value, double* diag, double* source)
{
    source[celli] += diag[celli] * value;
    diag[celli] += diag[celli];
}

// ---------------------------------------------------------------- vector (3-component) matrices
// fvMatrix<vector>: one scalar lduMatrix (diag/upper/lower) + vector source, psi and per-patch vector
// coefficients (AoS [n][3], as Field<vector>).  VectorSpaceI.H:376-418: cmptMax / cmptMin over the
// components, cmptAv = ((v0 + v1) + v2)/3.
__device__ __forceinline__ double cmpt_av3(const double* v) { return ((v[0] + v[1]) + v[2]) / 3; }

// fvMatrix.C:116-131, addBoundaryDiag(diag, cmpt)
__global__ void __launch_bounds__(GLUE_BLK)
glue_addBoundaryDiagCmpt_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                                const double* __restrict__ iC3, int cmpt, double* __restrict__ diag)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    const int b = cs[c], e = cs[c + 1];
    if (b == e) return;
    double d = diag[c];
    for (int j = b; j < e; j++) d += iC3[3 * cf[j] + cmpt];
    diag[c] = d;
}

// fvMatrix.C:150-178 for Type = vector (coupled: cmptMultiply(pbc, pnf))
__global__ void __launch_bounds__(GLUE_BLK)
glue_addBoundarySourceV_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                               const unsigned char* __restrict__ coupled, const double* __restrict__ bC3,
                               const double* __restrict__ pnf3, int couples, double* __restrict__ source3)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    const int b = cs[c], e = cs[c + 1];
    if (b == e) return;
    double s[3] = {source3[3 * c], source3[3 * c + 1], source3[3 * c + 2]};
    for (int j = b; j < e; j++)
    {
        const int f = cf[j];
        if (!coupled[f])
            for (int k = 0; k < 3; k++) s[k] += bC3[3 * f + k];
        else if (couples)
            for (int k = 0; k < 3; k++) s[k] += bC3[3 * f + k] * pnf3[3 * f + k];
    }
    for (int k = 0; k < 3; k++) source3[3 * c + k] = s[k];
}

// fvMatrix.C:525-655, Type = vector
__global__ void __launch_bounds__(GLUE_BLK)
glue_relaxV_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                   const unsigned char* __restrict__ coupled, const double* __restrict__ iC3,
                   const double* __restrict__ bC3, const int* __restrict__ losortStart,
                   const int* __restrict__ losort, const int* __restrict__ ownerStart,
                   const double* __restrict__ upper, const double* __restrict__ lower, double alpha,
                   const double* __restrict__ psi3, double* __restrict__ diag, double* __restrict__ source3)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    const double D0 = diag[c];
    double D = D0;
    double sumOff = 0.0;
    for (int j = losortStart[c]; j < losortStart[c + 1]; j++) sumOff += fabs(lower[losort[j]]);
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++) sumOff += fabs(upper[f]);
    const int b = cs[c], e = cs[c + 1];
    for (int j = b; j < e; j++)
    {
        const int f = cf[j];
        const double* ic = iC3 + 3 * f;
        if (coupled[f])
        {
            D += ic[0];                                   // component(iCoeffs[face], 0)
            sumOff += fabs(bC3[3 * f]);                   // mag(component(pCoeffs[face], 0))
        }
        else
            D += fmax(fmax(fabs(ic[0]), fabs(ic[1])), fabs(ic[2]));   // cmptMax(cmptMag(.))
    }
    D = fmax(fabs(D), sumOff);
    D /= alpha;
    for (int j = b; j < e; j++)
    {
        const int f = cf[j];
        const double* ic = iC3 + 3 * f;
        D -= coupled[f] ? ic[0] : fmin(fmin(ic[0], ic[1]), ic[2]);     // component 0 | cmptMin
    }
    diag[c] = D;
    const double dd = D - D0;
    for (int k = 0; k < 3; k++) source3[3 * c + k] += dd * psi3[3 * c + k];
}

// fvMatrix.C:722-746 with D() = diag + cmptAv(internalCoeffs) (:689-694)
__global__ void __launch_bounds__(GLUE_BLK)
glue_AV_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf, const double* __restrict__ iC3,
               const double* __restrict__ diag, const double* __restrict__ V, double* __restrict__ A)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    double d = diag[c];
    for (int j = cs[c]; j < cs[c + 1]; j++) d += cmpt_av3(iC3 + 3 * cf[j]);
    A[c] = d / V[c];
}

// fvMatrix<Type>::H, the generic template fvMatrix.C:751-813 (vector): per component
// (cmptAv(iC) - iC.component(cmpt)) summed over the cell's boundary faces, times psi; + lduMatrix::H(psi)
// + source; + boundary source; / V
__global__ void __launch_bounds__(GLUE_BLK)
glue_HV_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
               const unsigned char* __restrict__ coupled, const double* __restrict__ iC3,
               const double* __restrict__ bC3, const double* __restrict__ pnf3,
               const int* __restrict__ losortStart, const int* __restrict__ losort,
               const int* __restrict__ ownerStart, const int* __restrict__ l, const int* __restrict__ u,
               const double* __restrict__ upper, const double* __restrict__ lower,
               const double* __restrict__ psi3, const double* __restrict__ source3, const double* __restrict__ V,
               double* __restrict__ H3)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    const int b = cs[c], e = cs[c + 1];
    double h[3];
    for (int k = 0; k < 3; k++)
    {
        double bd = 0.0;                                  // addBoundaryDiag(bdc, cmpt)
        for (int j = b; j < e; j++) bd += iC3[3 * cf[j] + k];
        bd = -bd;                                         // negate
        for (int j = b; j < e; j++) bd += cmpt_av3(iC3 + 3 * cf[j]);   // addCmptAvBoundaryDiag
        h[k] = bd * psi3[3 * c + k];
    }
    double hl[3] = {0.0, 0.0, 0.0};                       // lduMatrix::H(psi), lduMatrixTemplates.C:32-69
    for (int j = losortStart[c]; j < losortStart[c + 1]; j++)
    {
        const int f = losort[j];
        for (int k = 0; k < 3; k++) hl[k] -= lower[f] * psi3[3 * l[f] + k];
    }
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
        for (int k = 0; k < 3; k++) hl[k] -= upper[f] * psi3[3 * u[f] + k];
    for (int k = 0; k < 3; k++) h[k] += hl[k] + source3[3 * c + k];
    for (int j = b; j < e; j++)
    {
        const int f = cf[j];
        for (int k = 0; k < 3; k++) h[k] += coupled[f] ? bC3[3 * f + k] * pnf3[3 * f + k] : bC3[3 * f + k];
    }
    for (int k = 0; k < 3; k++) H3[3 * c + k] = h[k] / V[c];
}

// ---------------------------------------------------------------- higher-order schemes
// linearUpwind<scalar>::correction, interpolation/surfaceInterpolation/schemes/linearUpwind/linearUpwind.C:87-91
// (internal faces): corr[f] = (Cf[f] - C[c]) & gradVf[c], c = faceFlux[f] > 0 ? owner : neighbour
__global__ void __launch_bounds__(GLUE_BLK)
fv_linearUpwind_kernel(int nFaces, const int* __restrict__ l, const int* __restrict__ u,
                       const double* __restrict__ phi, const double* __restrict__ C3,
                       const double* __restrict__ Cf3, const double* __restrict__ grad3, double* __restrict__ corr)
{
    const int f = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (f >= nFaces) return;
    const int c = phi[f] > 0 ? l[f] : u[f];
    const double dx = Cf3[3 * f] - C3[3 * c], dy = Cf3[3 * f + 1] - C3[3 * c + 1], dz = Cf3[3 * f + 2] - C3[3 * c + 2];
    corr[f] = dx * grad3[3 * c] + dy * grad3[3 * c + 1] + dz * grad3[3 * c + 2];
}

// cellLimitedGrad<scalar>::limitFace, cellLimitedGrad.H:136-152 (VSMALL = 1e-300, doubleScalar.H)
__device__ __forceinline__ void limit_face(double& limiter, double maxDelta, double minDelta, double extrapolate)
{
    if (extrapolate > maxDelta + 1.0e-300) limiter = fmin(limiter, maxDelta / extrapolate);
    else if (extrapolate < minDelta - 1.0e-300) limiter = fmin(limiter, minDelta / extrapolate);
}

// cellLimitedGrad<scalar>::calcGrad, gradSchemes/limitedGradSchemes/cellLimitedGrad/cellLimitedGrads.C:46-196:
// min / max of the neighbour values (internal faces, then patch faces: patchNeighbourField on coupled patches,
// the patch value otherwise), k-relaxation of the bounds, limiter = min over the cell's faces, g *= limiter.
// max / min are order-independent, so a per-cell gather reproduces the face loops exactly.
__global__ void __launch_bounds__(GLUE_BLK)
fv_cellLimitedGrad_kernel(int nCells, double k, const int* __restrict__ cs, const int* __restrict__ cf,
                          const int* __restrict__ losortStart, const int* __restrict__ losort,
                          const int* __restrict__ ownerStart, const int* __restrict__ l, const int* __restrict__ u,
                          const double* __restrict__ vsf, const double* __restrict__ bVal,
                          const double* __restrict__ C3, const double* __restrict__ Cf3,
                          const double* __restrict__ bCf3, double* __restrict__ g3)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    const double v = vsf[c];
    double mx = v, mn = v;
    for (int j = losortStart[c]; j < losortStart[c + 1]; j++)
    {
        const double o = vsf[l[losort[j]]];
        mx = fmax(mx, o); mn = fmin(mn, o);
    }
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
    {
        const double o = vsf[u[f]];
        mx = fmax(mx, o); mn = fmin(mn, o);
    }
    const int b = cs ? cs[c] : 0, e = cs ? cs[c + 1] : 0;
    for (int j = b; j < e; j++)
    {
        const double o = bVal[cf[j]];
        mx = fmax(mx, o); mn = fmin(mn, o);
    }
    mx -= v; mn -= v;
    if (k < 1.0)
    {
        const double mm = (1.0 / k - 1.0) * (mx - mn);
        mx += mm; mn -= mm;
    }
    const double gx = g3[3 * c], gy = g3[3 * c + 1], gz = g3[3 * c + 2];
    const double cx = C3[3 * c], cy = C3[3 * c + 1], cz = C3[3 * c + 2];
    double lim = 1.0;
    for (int j = losortStart[c]; j < losortStart[c + 1]; j++)
    {
        const int f = losort[j];
        limit_face(lim, mx, mn, (Cf3[3 * f] - cx) * gx + (Cf3[3 * f + 1] - cy) * gy + (Cf3[3 * f + 2] - cz) * gz);
    }
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
        limit_face(lim, mx, mn, (Cf3[3 * f] - cx) * gx + (Cf3[3 * f + 1] - cy) * gy + (Cf3[3 * f + 2] - cz) * gz);
    for (int j = b; j < e; j++)
    {
        const int f = cf[j];
        limit_face(lim, mx, mn, (bCf3[3 * f] - cx) * gx + (bCf3[3 * f + 1] - cy) * gy + (bCf3[3 * f + 2] - cz) * gz);
    }
    g3[3 * c] = gx * lim; g3[3 * c + 1] = gy * lim; g3[3 * c + 2] = gz * lim;
}

// vector & tensor (TensorI.H operator&(Vector, Tensor)): component j = v.x*t.xj + v.y*t.yj + v.z*t.zj
__device__ __forceinline__ void vec_dot_tensor(double dx, double dy, double dz, const double* __restrict__ t,
                                               double& ex, double& ey, double& ez)
{
    ex = dx * t[0] + dy * t[3] + dz * t[6];
    ey = dx * t[1] + dy * t[4] + dz * t[7];
    ez = dx * t[2] + dy * t[5] + dz * t[8];
}

// linearUpwindV<vector>::correction, schemes/linearUpwind/linearUpwindV.C:87-140 (internal faces): the upwind
// cell's gradient extrapolation, limited against the linear (central) correction maxCorr
__global__ void __launch_bounds__(GLUE_BLK)
fv_linearUpwindV_kernel(int nFaces, const int* __restrict__ l, const int* __restrict__ u,
                        const double* __restrict__ phi, const double* __restrict__ w, const double* __restrict__ vf3,
                        const double* __restrict__ C3, const double* __restrict__ Cf3,
                        const double* __restrict__ grad9, double* __restrict__ corr3)
{
    const int f = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (f >= nFaces) return;
    const int o = l[f], n = u[f];
    double mx, my, mz;
    int c;
    if (phi[f] > 0.0)
    {
        const double a = 1.0 - w[f];
        mx = a * (vf3[3 * n] - vf3[3 * o]); my = a * (vf3[3 * n + 1] - vf3[3 * o + 1]); mz = a * (vf3[3 * n + 2] - vf3[3 * o + 2]);
        c = o;
    }
    else
    {
        const double a = w[f];
        mx = a * (vf3[3 * o] - vf3[3 * n]); my = a * (vf3[3 * o + 1] - vf3[3 * n + 1]); mz = a * (vf3[3 * o + 2] - vf3[3 * n + 2]);
        c = n;
    }
    double sx, sy, sz;
    vec_dot_tensor(Cf3[3 * f] - C3[3 * c], Cf3[3 * f + 1] - C3[3 * c + 1], Cf3[3 * f + 2] - C3[3 * c + 2], grad9 + 9 * (size_t)c,
                   sx, sy, sz);
    const double sfCorrs = sx * sx + sy * sy + sz * sz;
    const double maxCorrs = sx * mx + sy * my + sz * mz;
    if (sfCorrs > 0)
    {
        if (maxCorrs < 0) { sx = 0.0; sy = 0.0; sz = 0.0; }
        else if (sfCorrs > maxCorrs)
        {
            const double r = maxCorrs / (sfCorrs + 1.0e-300);
            sx *= r; sy *= r; sz *= r;
        }
    }
    else if (sfCorrs < 0)
    {
        if (maxCorrs > 0) { sx = 0.0; sy = 0.0; sz = 0.0; }
        else if (sfCorrs < maxCorrs)
        {
            const double r = maxCorrs / (sfCorrs - 1.0e-300);
            sx *= r; sy *= r; sz *= r;
        }
    }
    corr3[3 * f] = sx; corr3[3 * f + 1] = sy; corr3[3 * f + 2] = sz;
}

// cellLimitedGrad<vector>::calcGrad, cellLimitedGrads.C:200-360: the scalar algorithm per component
// (limitFace, cellLimitedGrad.H:155-174); the limiter of component j scales column j of the gradient tensor
__global__ void __launch_bounds__(GLUE_BLK)
fv_cellLimitedGradV_kernel(int nCells, double k, const int* __restrict__ cs, const int* __restrict__ cf,
                           const int* __restrict__ losortStart, const int* __restrict__ losort,
                           const int* __restrict__ ownerStart, const int* __restrict__ l, const int* __restrict__ u,
                           const double* __restrict__ vsf3, const double* __restrict__ bVal3,
                           const double* __restrict__ C3, const double* __restrict__ Cf3,
                           const double* __restrict__ bCf3, double* __restrict__ g9)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    double v[3], mx[3], mn[3];
    for (int j = 0; j < 3; j++) { v[j] = vsf3[3 * c + j]; mx[j] = v[j]; mn[j] = v[j]; }
    for (int q = losortStart[c]; q < losortStart[c + 1]; q++)
    {
        const int o = l[losort[q]];
        for (int j = 0; j < 3; j++) { const double x = vsf3[3 * o + j]; mx[j] = fmax(mx[j], x); mn[j] = fmin(mn[j], x); }
    }
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
    {
        const int o = u[f];
        for (int j = 0; j < 3; j++) { const double x = vsf3[3 * o + j]; mx[j] = fmax(mx[j], x); mn[j] = fmin(mn[j], x); }
    }
    const int b = cs ? cs[c] : 0, e = cs ? cs[c + 1] : 0;
    for (int q = b; q < e; q++)
    {
        const int f = cf[q];
        for (int j = 0; j < 3; j++) { const double x = bVal3[3 * f + j]; mx[j] = fmax(mx[j], x); mn[j] = fmin(mn[j], x); }
    }
    for (int j = 0; j < 3; j++)
    {
        mx[j] -= v[j]; mn[j] -= v[j];
        if (k < 1.0)
        {
            const double mm = (1.0 / k - 1.0) * (mx[j] - mn[j]);
            mx[j] += mm; mn[j] -= mm;
        }
    }
    double t[9];
    for (int j = 0; j < 9; j++) t[j] = g9[9 * (size_t)c + j];
    const double cx = C3[3 * c], cy = C3[3 * c + 1], cz = C3[3 * c + 2];
    double lim[3] = {1.0, 1.0, 1.0};
    double ex[3];
    for (int q = losortStart[c]; q < losortStart[c + 1]; q++)
    {
        const int f = losort[q];
        vec_dot_tensor(Cf3[3 * f] - cx, Cf3[3 * f + 1] - cy, Cf3[3 * f + 2] - cz, t, ex[0], ex[1], ex[2]);
        for (int j = 0; j < 3; j++) limit_face(lim[j], mx[j], mn[j], ex[j]);
    }
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
    {
        vec_dot_tensor(Cf3[3 * f] - cx, Cf3[3 * f + 1] - cy, Cf3[3 * f + 2] - cz, t, ex[0], ex[1], ex[2]);
        for (int j = 0; j < 3; j++) limit_face(lim[j], mx[j], mn[j], ex[j]);
    }
    for (int q = b; q < e; q++)
    {
        const int f = cf[q];
        vec_dot_tensor(bCf3[3 * f] - cx, bCf3[3 * f + 1] - cy, bCf3[3 * f + 2] - cz, t, ex[0], ex[1], ex[2]);
        for (int j = 0; j < 3; j++) limit_face(lim[j], mx[j], mn[j], ex[j]);
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) g9[9 * (size_t)c + 3 * i + j] = lim[j] * t[3 * i + j];
}

// fv::gaussGrad<Type>::gradf (gaussGrad.C:41-110) with the patch faces: grad = (sum_f +-Sf*ssf_f + sum_b Sf_b*ssf_b)/V,
// Sf*ssf the outer product for a vector field (tensor component 3i+j = Sf_i*ssf_j).  Per cell and component the
// products are added in the reference's order: neighbour faces (-=) and owned faces (+=) in face order, then the
// patch faces in (patch, face) order.
template <int NC>
__global__ void __launch_bounds__(GLUE_BLK)
fv_gaussGradFull_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                        const int* __restrict__ losortStart, const int* __restrict__ losort,
                        const int* __restrict__ ownerStart, const double* __restrict__ Sf3,
                        const double* __restrict__ ssf, const double* __restrict__ bSf3,
                        const double* __restrict__ bssf, const double* __restrict__ V, double* __restrict__ grad)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    double acc[3 * NC];
#pragma unroll
    for (int q = 0; q < 3 * NC; q++) acc[q] = 0.0;
    for (int t = losortStart[c]; t < losortStart[c + 1]; t++)
    {
        const int f = losort[t];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < NC; j++) acc[NC * i + j] -= Sf3[3 * (size_t)f + i] * ssf[NC * (size_t)f + j];
    }
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < NC; j++) acc[NC * i + j] += Sf3[3 * (size_t)f + i] * ssf[NC * (size_t)f + j];
    }
    if (cs)
        for (int q = cs[c]; q < cs[c + 1]; q++)
        {
            const int f = cf[q];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < NC; j++) acc[NC * i + j] += bSf3[3 * (size_t)f + i] * bssf[NC * (size_t)f + j];
        }
    const double v = V[c];
#pragma unroll
    for (int q = 0; q < 3 * NC; q++) grad[3 * NC * (size_t)c + q] = acc[q] / v;
}

// The same sums with the owned faces of a tile of GLUE_BLK consecutive cells staged through LDS (north_star: "LDS-staged
// face -> cell scatter"): faces are ordered by owner, so the tile owns ONE contiguous face range; its products Sf*ssf are
// formed once by coalesced loads (a lane per (face, component)) and kept in LDS, and every cell then takes its owned faces -
// and those neighbour-side faces that the tile owns too - from there.  Only the neighbour-side faces owned by other
// tiles are gathered per lane from global memory (216^3 box: 2 of a cell's 6 faces instead of 6).  Same products, same
// order of additions per cell and component as fv_gaussGradFull_kernel.
// Workgroups go to the eight XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own L2: with tile =
// workgroup index, neighbouring tiles - which read each other's faces - never share an L2.  Here XCD x walks the x-th
// eighth of the tiles in order, so the faces a tile gathers from its predecessors were staged by the same XCD moments ago.
__device__ __forceinline__ int xcd_tile(int nTiles)
{
    const int per = (nTiles + 7) >> 3;
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}
static inline int xcd_tile_grid(int nCells) { const int nT = (nCells + GLUE_BLK - 1) / GLUE_BLK; return 8 * ((nT + 7) / 8); }
#define GG_MAXF 768   // staged faces per tile (x 3*NC doubles of LDS); a tile owning more stages the first GG_MAXF
template <int NC>
__global__ void __launch_bounds__(GLUE_BLK)
fv_gaussGradTile_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                        const int* __restrict__ losortStart, const int* __restrict__ losort,
                        const int* __restrict__ ownerStart, const double* __restrict__ Sf3,
                        const double* __restrict__ ssf, const double* __restrict__ bSf3,
                        const double* __restrict__ bssf, const double* __restrict__ V, double* __restrict__ grad)
{
    constexpr int K = 3 * NC;
    __shared__ double prod[GG_MAXF * K];
    const int nTiles = (nCells + GLUE_BLK - 1) / GLUE_BLK;
    const int tile = xcd_tile(nTiles);
    if (tile >= nTiles) return;
    const int c0 = tile * GLUE_BLK;
    const int cEnd = c0 + GLUE_BLK < nCells ? c0 + GLUE_BLK : nCells;
    const int fA = ownerStart[c0];
    int nOwn = ownerStart[cEnd] - fA;
    if (nOwn > GG_MAXF) nOwn = GG_MAXF;
    for (int e = threadIdx.x; e < nOwn * K; e += GLUE_BLK)
    {
        const int fl = e / K, q = e - fl * K;
        const int i = q / NC, j = q - i * NC;
        const size_t f = (size_t)(fA + fl);
        prod[e] = Sf3[3 * f + i] * ssf[NC * f + j];
    }
    __syncthreads();
    const int c = c0 + threadIdx.x;
    if (c >= nCells) return;
    double acc[K];
#pragma unroll
    for (int q = 0; q < K; q++) acc[q] = 0.0;
    for (int t = losortStart[c]; t < losortStart[c + 1]; t++)
    {
        const int f = losort[t];
        const unsigned fl = (unsigned)(f - fA);
        if (fl < (unsigned)nOwn)
        {
#pragma unroll
            for (int q = 0; q < K; q++) acc[q] -= prod[fl * K + q];
        }
        else
        {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < NC; j++) acc[NC * i + j] -= Sf3[3 * (size_t)f + i] * ssf[NC * (size_t)f + j];
        }
    }
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
    {
        const unsigned fl = (unsigned)(f - fA);
        if (fl < (unsigned)nOwn)
        {
#pragma unroll
            for (int q = 0; q < K; q++) acc[q] += prod[fl * K + q];
        }
        else
        {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < NC; j++) acc[NC * i + j] += Sf3[3 * (size_t)f + i] * ssf[NC * (size_t)f + j];
        }
    }
    if (cs)
        for (int q = cs[c]; q < cs[c + 1]; q++)
        {
            const int f = cf[q];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < NC; j++) acc[NC * i + j] += bSf3[3 * (size_t)f + i] * bssf[NC * (size_t)f + j];
        }
    const double v = V[c];
#pragma unroll
    for (int q = 0; q < K; q++) grad[K * (size_t)c + q] = acc[q] / v;
}

// Vector field (tensor gradient): the tile's owned faces are staged RAW - Sf and ssf, 6 doubles per face, copied from the
// two contiguous global ranges by coalesced loads (the per-lane loads of fv_gaussGradFull_kernel<3> walk 24-byte records:
// nine load instructions of 64 x 8 bytes spread over 4.6 KB each, the texture path of the CU is the bound, not HBM) -
// and every cell forms its products from LDS; 9 products per face in LDS (fv_gaussGradTile_kernel<3>) cost 55 KB per
// workgroup and measured slower.  The gradient leaves through LDS as well: 72-byte records per lane become coalesced
// rows.  Same products, same order of additions per cell and component as fv_gaussGradFull_kernel.
__global__ void __launch_bounds__(GLUE_BLK)
fv_gaussGradTile3_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                         const int* __restrict__ losortStart, const int* __restrict__ losort,
                         const int* __restrict__ ownerStart, const double* __restrict__ Sf3,
                         const double* __restrict__ ssf, const double* __restrict__ bSf3,
                         const double* __restrict__ bssf, const double* __restrict__ V, double* __restrict__ grad)
{
    static_assert(GG_MAXF * 3 >= GLUE_BLK * 9, "the staging buffer of Sf doubles as the output tile");
    __shared__ double sS[GG_MAXF * 3];
    __shared__ double sV[GG_MAXF * 3];
    const int nTiles = (nCells + GLUE_BLK - 1) / GLUE_BLK;
    const int tile = xcd_tile(nTiles);
    if (tile >= nTiles) return;
    const int c0 = tile * GLUE_BLK;
    const int cEnd = c0 + GLUE_BLK < nCells ? c0 + GLUE_BLK : nCells;
    const int fA = ownerStart[c0];
    int nOwn = ownerStart[cEnd] - fA;
    if (nOwn > GG_MAXF) nOwn = GG_MAXF;
    for (int e = threadIdx.x; e < nOwn * 3; e += GLUE_BLK)
    {
        sS[e] = Sf3[3 * (size_t)fA + e];
        sV[e] = ssf[3 * (size_t)fA + e];
    }
    __syncthreads();
    const int c = c0 + threadIdx.x;
    const bool have = c < nCells;
    double acc[9];
#pragma unroll
    for (int q = 0; q < 9; q++) acc[q] = 0.0;
    if (have)
    {
        // faces four at a time: their indices, then their six values each (LDS or global), in flight together; the sums
        // stay in face order
        const int t1 = losortStart[c + 1];
        for (int t = losortStart[c]; t < t1; t += 4)
        {
            int f[4];
            double S[4][3], v[4][3];
#pragma unroll
            for (int k = 0; k < 4; k++) f[k] = losort[t + k < t1 ? t + k : t1 - 1];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const unsigned fl = (unsigned)(f[k] - fA);
                if (fl < (unsigned)nOwn)
                {
#pragma unroll
                    for (int i = 0; i < 3; i++) { S[k][i] = sS[3 * fl + i]; v[k][i] = sV[3 * fl + i]; }
                }
                else
                {
#pragma unroll
                    for (int i = 0; i < 3; i++) { S[k][i] = Sf3[3 * (size_t)f[k] + i]; v[k][i] = ssf[3 * (size_t)f[k] + i]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (t + k < t1)
                {
#pragma unroll
                    for (int i = 0; i < 3; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++) acc[3 * i + j] -= S[k][i] * v[k][j];
                }
        }
        for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
        {
            const unsigned fl = (unsigned)(f - fA);
            double S[3], v[3];
            if (fl < (unsigned)nOwn)
            {
#pragma unroll
                for (int i = 0; i < 3; i++) { S[i] = sS[3 * fl + i]; v[i] = sV[3 * fl + i]; }
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 3; i++) { S[i] = Sf3[3 * (size_t)f + i]; v[i] = ssf[3 * (size_t)f + i]; }
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) acc[3 * i + j] += S[i] * v[j];
        }
        if (cs)
            for (int q = cs[c]; q < cs[c + 1]; q++)
            {
                const int f = cf[q];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) acc[3 * i + j] += bSf3[3 * (size_t)f + i] * bssf[3 * (size_t)f + j];
            }
        const double vol = V[c];
#pragma unroll
        for (int q = 0; q < 9; q++) acc[q] = acc[q] / vol;
    }
    __syncthreads();      // every lane is done with the staged faces: sS becomes the output tile
    if (have)
    {
#pragma unroll
        for (int q = 0; q < 9; q++) sS[9 * threadIdx.x + q] = acc[q];
    }
    __syncthreads();
    const int nOut = (cEnd - c0) * 9;
    for (int e = threadIdx.x; e < nOut; e += GLUE_BLK) grad[9 * (size_t)c0 + e] = sS[e];
}

// the `bounded` convection wrapper (boundedConvectionScheme.C:60-77): diag -= V * surfaceIntegrate(phi),
// surfaceIntegrate as fvcSurfaceIntegrate.C:43-76 - the cell's neighbour faces (-=) and owned faces (+=) in face
// order, then its patch faces in (patch, face) order, then / V
__global__ void __launch_bounds__(GLUE_BLK)
fv_boundedSp_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                    const int* __restrict__ losortStart, const int* __restrict__ losort,
                    const int* __restrict__ ownerStart, const double* __restrict__ phi,
                    const double* __restrict__ bPhi, const double* __restrict__ V, double* __restrict__ diag)
{
    const int c = blockIdx.x * GLUE_BLK + threadIdx.x;
    if (c >= nCells) return;
    double acc = 0.0;
    for (int t = losortStart[c]; t < losortStart[c + 1]; t++) acc -= phi[losort[t]];
    for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++) acc += phi[f];
    if (cs)
        for (int j = cs[c]; j < cs[c + 1]; j++) acc += bPhi[cf[j]];
    const double v = V[c];
    diag[c] -= v * (acc / v);
}

// ---------------------------------------------------------------- C ABI
extern "C" {

int ldu_fv_boundary_create(ldu_addr* a, int32_t nPatches, const int32_t* patchSizes, const int32_t* faceCells,
                           const int32_t* coupled, ldu_fv_boundary** out)
{
    if (!a || !out || nPatches < 0) { ldu_set_error("ldu_fv_boundary_create: bad arguments"); return -2; }
    ldu_fv_boundary* b = new ldu_fv_boundary();
    b->a = a;
    b->nPatches = nPatches;
    int off = 0;
    for (int p = 0; p < nPatches; p++)
    {
        b->sizes.push_back(patchSizes[p]);
        b->offsets.push_back(off);
        b->coupled.push_back(coupled ? coupled[p] : 0);
        off += patchSizes[p];
    }
    b->nFacesTotal = off;
    const int nC = a->nCells;
    std::vector<int> cs(nC + 1, 0), cf(off), fc(off);
    std::vector<unsigned char> cp(off, 0);
    for (int i = 0; i < off; i++)
    {
        if (faceCells[i] < 0 || faceCells[i] >= nC)
        {
            delete b;
            ldu_set_error("ldu_fv_boundary_create: faceCells out of range");
            return -2;
        }
        fc[i] = faceCells[i];
        cs[faceCells[i] + 1]++;
    }
    for (int c = 0; c < nC; c++) cs[c + 1] += cs[c];
    {
        std::vector<int> pos(cs.begin(), cs.end() - 1);
        for (int p = 0; p < nPatches; p++)
            for (int i = 0; i < b->sizes[p]; i++)
            {
                const int g = b->offsets[p] + i;           // ascending (patch, face): the reference's order
                cf[pos[fc[g]]++] = g;
                cp[g] = (unsigned char)(b->coupled[p] ? 1 : 0);
            }
    }
    auto up = [&](void** d, const void* h, size_t bytes) -> int {
        LDU_CHECK_HIP(hipMalloc(d, bytes ? bytes : 4));
        if (bytes) LDU_CHECK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    if (up((void**)&b->d_cellStart, cs.data(), sizeof(int) * cs.size())
        || up((void**)&b->d_cellFace, cf.data(), sizeof(int) * cf.size())
        || up((void**)&b->d_faceCells, fc.data(), sizeof(int) * fc.size())
        || up((void**)&b->d_coupled, cp.data(), cp.size()))
    {
        delete b;
        return -1;
    }
    *out = b;
    return 0;
}

int ldu_fv_boundary_destroy(ldu_fv_boundary* b)
{
    if (!b) return 0;
    (void)hipFree(b->d_cellStart); (void)hipFree(b->d_cellFace);
    (void)hipFree(b->d_faceCells); (void)hipFree(b->d_coupled);
    delete b;
    return 0;
}

int ldu_fvm_addBoundaryDiag(ldu_fv_boundary* b, const double* internalCoeffs, double* diag)
{
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs, b->nFacesTotal);
    double* d = B.inout(diag, a->nCells, true);
    glue_addBoundaryDiag_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart, b->d_cellFace,
                                                                         iC, d);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(diag, d, a->nCells);
}

int ldu_fvm_addBoundarySource(ldu_fv_boundary* b, const double* boundaryCoeffs, const double* patchNeighbourField,
                              int32_t couples, double* source)
{
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* bC = B.in(boundaryCoeffs, b->nFacesTotal);
    const double* pnf = B.in(patchNeighbourField, b->nFacesTotal);
    if (couples && !pnf)
        for (int c : b->coupled)
            if (c) { ldu_set_error("ldu_fvm_addBoundarySource: coupled patches need patchNeighbourField"); return -2; }
    double* s = B.inout(source, a->nCells, true);
    glue_addBoundarySource_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart,
        b->d_cellFace, b->d_coupled, bC, pnf, couples, s);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(source, s, a->nCells);
}

int ldu_fvm_relax(ldu_fv_boundary* b, double alpha, const double* internalCoeffs, const double* boundaryCoeffs,
                  const double* upper, const double* lower, const double* psi, double* diag, double* source)
{
    if (alpha <= 0) return 0;   // fvMatrix.C:527-530
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs, b->nFacesTotal);
    const double* bC = B.in(boundaryCoeffs, b->nFacesTotal);
    const double* up = B.in(upper, a->nFaces);
    const double* lo = lower ? B.in(lower, a->nFaces) : up;
    const double* x = B.in(psi, a->nCells);
    double* d = B.inout(diag, a->nCells, true);
    double* s = B.inout(source, a->nCells, true);
    glue_relax_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart, b->d_cellFace,
        b->d_coupled, iC, bC, a->d_losortStart, a->d_losort, a->d_ownerStart, up, lo, alpha, x, d, s);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(diag, d, a->nCells)) return -1;
    return B.finish(source, s, a->nCells);
}

int ldu_fvm_setReference(ldu_addr* a, int32_t celli, double value, double* diag, double* source)
{
    if (celli < 0) return 0;    // fvMatrix.C:516
    if (celli >= a->nCells) { ldu_set_error("ldu_fvm_setReference: cell out of range"); return -2; }
    GlueBuf B(a->ctx->stream);
    double* d = B.inout(diag, a->nCells, true);
    double* s = B.inout(source, a->nCells, true);
    glue_setReference_kernel<<<1, 1, 0, B.s>>>(celli, value, d, s);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(diag, d, a->nCells)) return -1;
    return B.finish(source, s, a->nCells);
}

int ldu_fvm_A(ldu_fv_boundary* b, const double* internalCoeffs, const double* diag, const double* V, double* A)
{
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs, b->nFacesTotal);
    const double* d = B.in(diag, a->nCells);
    const double* v = B.in(V, a->nCells);
    double* o = B.inout(A, a->nCells, false);
    glue_A_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart, b->d_cellFace, iC, d, v, o);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(A, o, a->nCells);
}

int ldu_fvm_H(ldu_fv_boundary* b, const double* internalCoeffs, const double* boundaryCoeffs,
              const double* patchNeighbourField, const double* upper, const double* lower, const double* psi,
              const double* source, const double* V, double* H)
{
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs, b->nFacesTotal);
    const double* bC = B.in(boundaryCoeffs, b->nFacesTotal);
    const double* pnf = B.in(patchNeighbourField, b->nFacesTotal);
    const double* up = B.in(upper, a->nFaces);
    const double* lo = lower ? B.in(lower, a->nFaces) : up;
    const double* x = B.in(psi, a->nCells);
    const double* s = B.in(source, a->nCells);
    const double* v = B.in(V, a->nCells);
    double* o = B.inout(H, a->nCells, false);
    if (!pnf) pnf = bC;   // never dereferenced without coupled patches
    {
        const int nTiles = (a->nCells + GLUE_BLK - 1) / GLUE_BLK;
        glue_H_tile_kernel<<<8 * ((nTiles + 7) / 8), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart, b->d_cellFace, b->d_coupled,
            bC, pnf, a->d_losortStart, a->d_losort, a->d_ownerStart, a->d_l, a->d_u, up, lo, x, s, v, o);
    }
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(H, o, a->nCells);
}

int ldu_fvm_flux(ldu_fv_boundary* b, const double* internalCoeffs, const double* boundaryCoeffs,
                 const double* patchNeighbourField, const double* upper, const double* lower, const double* psi,
                 double* fluxInternal, double* fluxBoundary)
{
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs, b->nFacesTotal);
    const double* bC = B.in(boundaryCoeffs, b->nFacesTotal);
    const double* pnf = B.in(patchNeighbourField, b->nFacesTotal);
    const double* up = B.in(upper, a->nFaces);
    const double* lo = lower ? B.in(lower, a->nFaces) : up;
    const double* x = B.in(psi, a->nCells);
    double* fi = B.inout(fluxInternal, a->nFaces, false);
    double* fb = B.inout(fluxBoundary, b->nFacesTotal, false);
    if (!pnf) pnf = bC;
    glue_flux_internal_kernel<<<glue_grid(a->nFaces), GLUE_BLK, 0, B.s>>>(a->nFaces, a->d_l, a->d_u, up, lo, x, fi);
    if (b->nFacesTotal)
        glue_flux_boundary_kernel<<<glue_grid(b->nFacesTotal), GLUE_BLK, 0, B.s>>>(b->nFacesTotal, b->d_faceCells,
            b->d_coupled, iC, bC, pnf, x, fb);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(fluxInternal, fi, a->nFaces)) return -1;
    return B.finish(fluxBoundary, fb, b->nFacesTotal);
}

int ldu_fvm_addBoundaryDiagCmpt(ldu_fv_boundary* b, const double* internalCoeffs3, int32_t cmpt, double* diag)
{
    if (cmpt < 0 || cmpt > 2) { ldu_set_error("ldu_fvm_addBoundaryDiagCmpt: cmpt must be 0..2"); return -2; }
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs3, 3 * (size_t)b->nFacesTotal);
    double* d = B.inout(diag, a->nCells, true);
    glue_addBoundaryDiagCmpt_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart,
        b->d_cellFace, iC, cmpt, d);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(diag, d, a->nCells);
}

int ldu_fvm_addBoundarySourceV(ldu_fv_boundary* b, const double* boundaryCoeffs3, const double* patchNeighbourField3,
                               int32_t couples, double* source3)
{
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* bC = B.in(boundaryCoeffs3, 3 * (size_t)b->nFacesTotal);
    const double* pnf = B.in(patchNeighbourField3, 3 * (size_t)b->nFacesTotal);
    if (!pnf) pnf = bC;
    double* s = B.inout(source3, 3 * (size_t)a->nCells, true);
    glue_addBoundarySourceV_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart,
        b->d_cellFace, b->d_coupled, bC, pnf, couples, s);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(source3, s, 3 * (size_t)a->nCells);
}

int ldu_fvm_relaxV(ldu_fv_boundary* b, double alpha, const double* internalCoeffs3, const double* boundaryCoeffs3,
                   const double* upper, const double* lower, const double* psi3, double* diag, double* source3)
{
    if (alpha <= 0) return 0;
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs3, 3 * (size_t)b->nFacesTotal);
    const double* bC = B.in(boundaryCoeffs3, 3 * (size_t)b->nFacesTotal);
    const double* up = B.in(upper, a->nFaces);
    const double* lo = lower ? B.in(lower, a->nFaces) : up;
    const double* x = B.in(psi3, 3 * (size_t)a->nCells);
    double* d = B.inout(diag, a->nCells, true);
    double* s = B.inout(source3, 3 * (size_t)a->nCells, true);
    glue_relaxV_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart, b->d_cellFace,
        b->d_coupled, iC, bC, a->d_losortStart, a->d_losort, a->d_ownerStart, up, lo, alpha, x, d, s);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(diag, d, a->nCells)) return -1;
    return B.finish(source3, s, 3 * (size_t)a->nCells);
}

int ldu_fvm_AV(ldu_fv_boundary* b, const double* internalCoeffs3, const double* diag, const double* V, double* A)
{
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs3, 3 * (size_t)b->nFacesTotal);
    const double* d = B.in(diag, a->nCells);
    const double* v = B.in(V, a->nCells);
    double* o = B.inout(A, a->nCells, false);
    glue_AV_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart, b->d_cellFace, iC, d, v, o);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(A, o, a->nCells);
}

int ldu_fvm_HV(ldu_fv_boundary* b, const double* internalCoeffs3, const double* boundaryCoeffs3,
               const double* patchNeighbourField3, const double* upper, const double* lower, const double* psi3,
               const double* source3, const double* V, double* H3)
{
    ldu_addr* a = b->a;
    GlueBuf B(a->ctx->stream);
    const double* iC = B.in(internalCoeffs3, 3 * (size_t)b->nFacesTotal);
    const double* bC = B.in(boundaryCoeffs3, 3 * (size_t)b->nFacesTotal);
    const double* pnf = B.in(patchNeighbourField3, 3 * (size_t)b->nFacesTotal);
    const double* up = B.in(upper, a->nFaces);
    const double* lo = lower ? B.in(lower, a->nFaces) : up;
    const double* x = B.in(psi3, 3 * (size_t)a->nCells);
    const double* s = B.in(source3, 3 * (size_t)a->nCells);
    const double* v = B.in(V, a->nCells);
    double* o = B.inout(H3, 3 * (size_t)a->nCells, false);
    if (!pnf) pnf = bC;
    glue_HV_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b->d_cellStart, b->d_cellFace, b->d_coupled,
        iC, bC, pnf, a->d_losortStart, a->d_losort, a->d_ownerStart, a->d_l, a->d_u, up, lo, x, s, v, o);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(H3, o, 3 * (size_t)a->nCells);
}

int ldu_fv_linearUpwindCorrection(ldu_addr* a, const double* faceFlux, const double* C3, const double* Cf3,
                                  const double* gradVf3, double* corr)
{
    GlueBuf B(a->ctx->stream);
    const double* phi = B.in(faceFlux, a->nFaces);
    const double* c = B.in(C3, 3 * (size_t)a->nCells);
    const double* cf = B.in(Cf3, 3 * (size_t)a->nFaces);
    const double* g = B.in(gradVf3, 3 * (size_t)a->nCells);
    double* o = B.inout(corr, a->nFaces, false);
    fv_linearUpwind_kernel<<<glue_grid(a->nFaces), GLUE_BLK, 0, B.s>>>(a->nFaces, a->d_l, a->d_u, phi, c, cf, g, o);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(corr, o, a->nFaces);
}

int ldu_fvc_cellLimitedGrad(ldu_addr* a, ldu_fv_boundary* b, double k, const double* vsf, const double* boundaryValues,
                            const double* C3, const double* Cf3, const double* boundaryCf3, double* grad3)
{
    if (k < 1.0e-15) return 0;   // cellLimitedGrads.C:58-61 (k_ < SMALL: unlimited)
    if (b && b->a != a) { ldu_set_error("ldu_fvc_cellLimitedGrad: boundary belongs to another addressing"); return -2; }
    GlueBuf B(a->ctx->stream);
    const size_t nB = b ? (size_t)b->nFacesTotal : 0;
    const double* v = B.in(vsf, a->nCells);
    const double* bv = B.in(boundaryValues, nB);
    const double* c = B.in(C3, 3 * (size_t)a->nCells);
    const double* cf = B.in(Cf3, 3 * (size_t)a->nFaces);
    const double* bcf = B.in(boundaryCf3, 3 * nB);
    double* g = B.inout(grad3, 3 * (size_t)a->nCells, true);
    fv_cellLimitedGrad_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, k, b ? b->d_cellStart : nullptr,
        b ? b->d_cellFace : nullptr, a->d_losortStart, a->d_losort, a->d_ownerStart, a->d_l, a->d_u, v, bv, c, cf,
        bcf, g);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(grad3, g, 3 * (size_t)a->nCells);
}

int ldu_fvc_gaussGradFull(ldu_addr* a, ldu_fv_boundary* b, int32_t nComp, const double* Sf3, const double* ssf,
                          const double* boundarySf3, const double* boundarySsf, const double* V, double* grad)
{
    if (nComp != 1 && nComp != 3) { ldu_set_error("ldu_fvc_gaussGradFull: nComp must be 1 or 3"); return -2; }
    if (b && b->a != a) { ldu_set_error("ldu_fvc_gaussGradFull: boundary belongs to another addressing"); return -2; }
    GlueBuf B(a->ctx->stream);
    const size_t nB = b ? (size_t)b->nFacesTotal : 0;
    const double* sf = B.in(Sf3, 3 * (size_t)a->nFaces);
    const double* f = B.in(ssf, (size_t)nComp * a->nFaces);
    const double* bsf = B.in(boundarySf3, 3 * nB);
    const double* bf = B.in(boundarySsf, (size_t)nComp * nB);
    const double* v = B.in(V, a->nCells);
    double* g = B.inout(grad, 3 * (size_t)nComp * a->nCells, false);
    if (nComp == 1)
        fv_gaussGradTile_kernel<1><<<xcd_tile_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b ? b->d_cellStart : nullptr,
            b ? b->d_cellFace : nullptr, a->d_losortStart, a->d_losort, a->d_ownerStart, sf, f, bsf, bf, v, g);
    else
        fv_gaussGradTile3_kernel<<<xcd_tile_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b ? b->d_cellStart : nullptr,
            b ? b->d_cellFace : nullptr, a->d_losortStart, a->d_losort, a->d_ownerStart, sf, f, bsf, bf, v, g);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(grad, g, 3 * (size_t)nComp * a->nCells);
}

int ldu_fvm_boundedSp(ldu_addr* a, ldu_fv_boundary* b, const double* faceFlux, const double* boundaryFlux,
                      const double* V, double* diag)
{
    if (b && b->a != a) { ldu_set_error("ldu_fvm_boundedSp: boundary belongs to another addressing"); return -2; }
    GlueBuf B(a->ctx->stream);
    const size_t nB = b ? (size_t)b->nFacesTotal : 0;
    const double* phi = B.in(faceFlux, a->nFaces);
    const double* bphi = B.in(boundaryFlux, nB);
    const double* v = B.in(V, a->nCells);
    double* d = B.inout(diag, a->nCells, true);
    fv_boundedSp_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, b ? b->d_cellStart : nullptr,
        b ? b->d_cellFace : nullptr, a->d_losortStart, a->d_losort, a->d_ownerStart, phi, bphi, v, d);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(diag, d, a->nCells);
}

int ldu_fv_linearUpwindVCorrection(ldu_addr* a, const double* faceFlux, const double* weights, const double* vf3,
                                   const double* C3, const double* Cf3, const double* gradVf9, double* corr3)
{
    GlueBuf B(a->ctx->stream);
    const double* phi = B.in(faceFlux, a->nFaces);
    const double* w = B.in(weights, a->nFaces);
    const double* v = B.in(vf3, 3 * (size_t)a->nCells);
    const double* c = B.in(C3, 3 * (size_t)a->nCells);
    const double* cf = B.in(Cf3, 3 * (size_t)a->nFaces);
    const double* g = B.in(gradVf9, 9 * (size_t)a->nCells);
    double* o = B.inout(corr3, 3 * (size_t)a->nFaces, false);
    fv_linearUpwindV_kernel<<<glue_grid(a->nFaces), GLUE_BLK, 0, B.s>>>(a->nFaces, a->d_l, a->d_u, phi, w, v, c, cf, g, o);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(corr3, o, 3 * (size_t)a->nFaces);
}

int ldu_fvc_cellLimitedGradV(ldu_addr* a, ldu_fv_boundary* b, double k, const double* vsf3, const double* boundaryValues3,
                             const double* C3, const double* Cf3, const double* boundaryCf3, double* grad9)
{
    if (k < 1.0e-15) return 0;   // cellLimitedGrads.C:212-215 (k_ < SMALL: unlimited)
    if (b && b->a != a) { ldu_set_error("ldu_fvc_cellLimitedGradV: boundary belongs to another addressing"); return -2; }
    GlueBuf B(a->ctx->stream);
    const size_t nB = b ? (size_t)b->nFacesTotal : 0;
    const double* v = B.in(vsf3, 3 * (size_t)a->nCells);
    const double* bv = B.in(boundaryValues3, 3 * nB);
    const double* c = B.in(C3, 3 * (size_t)a->nCells);
    const double* cf = B.in(Cf3, 3 * (size_t)a->nFaces);
    const double* bcf = B.in(boundaryCf3, 3 * nB);
    double* g = B.inout(grad9, 9 * (size_t)a->nCells, true);
    fv_cellLimitedGradV_kernel<<<glue_grid(a->nCells), GLUE_BLK, 0, B.s>>>(a->nCells, k, b ? b->d_cellStart : nullptr,
        b ? b->d_cellFace : nullptr, a->d_losortStart, a->d_losort, a->d_ownerStart, a->d_l, a->d_u, v, bv, c, cf,
        bcf, g);
    LDU_CHECK_HIP(hipGetLastError());
    return B.finish(grad9, g, 9 * (size_t)a->nCells);
}

}  // extern "C"
