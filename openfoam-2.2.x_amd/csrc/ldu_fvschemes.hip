// Non-orthogonal correction, gaussDiv and the patch halves of interpolate / gaussGrad:
// what motorBike's fvSchemes (`laplacianSchemes default Gauss linear corrected; snGradSchemes default corrected;
// div((nuEff*dev(T(grad(U))))) Gauss linear;`) adds to the uncorrected stencils of ldu_kernels.hip / ldu_fvmatrix.hip.
//
// Reference (src/finiteVolume): interpolation/surfaceInterpolation/surfaceInterpolation/surfaceInterpolation.C:252-391
// (nonOrthDeltaCoeffs, nonOrthCorrectionVectors), finiteVolume/snGradSchemes/correctedSnGrad/correctedSnGrad.C:44-107,
// snGradScheme/snGradScheme.C:104-186, laplacianSchemes/gaussLaplacianScheme/gaussLaplacianSchemes.C:43-114 and
// gaussLaplacianScheme.C:92-231 (tensor diffusivity), divSchemes/gaussDivScheme/gaussDivScheme.C:48-68,
// surfaceInterpolationScheme.C:298-314 (patch faces), gradSchemes/gaussGrad/gaussGrad.C:144-170.
//
// Layout: ORIGINAL cell / face numbering, vectors [n][3], tensors [n][9] (xx xy xz yx yy yz zx zy zz), symmTensors
// [n][6] (xx xy xz yy yz zz), patch-face arrays concatenated in ldu_fv_boundary order.  Face kernels are one thread
// per face; cell kernels gather a cell's faces in the order the reference's face loops reach it (faces whose upper
// cell it is, ascending; the faces it owns, ascending; its patch faces in (patch, face) order).  Every expression is
// evaluated in the reference's operation order with -ffp-contract=off: results are bit-identical.
#include <vector>

#include "ldu_internal.hpp"

#define FS_BLK 256
static inline int fs_grid(long n) { long g = (n + FS_BLK - 1) / FS_BLK; return g < 1 ? 1 : (g > 65535 ? 65535 : (int)g); }
static const double kVSmall = 1e-300;

static bool fs_dev_ptr(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice;
}

struct FsBuf {
    hipStream_t s;
    std::vector<void*> owned;
    bool bad = false;   // an allocation or upload failed: nothing may be launched on these pointers
    explicit FsBuf(hipStream_t ss) : s(ss) {}
    int failed() const
    {
        if (bad) ldu_set_error("fv schemes: device allocation or upload of an argument failed");
        return bad ? -1 : 0;
    }
    ~FsBuf() { for (void* p : owned) (void)hipFree(p); }
    template <class T>
    const T* in(const T* user, size_t n)
    {
        if (!user || fs_dev_ptr(user)) return user;
        T* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(T) * (n ? n : 1)) != hipSuccess) { bad = true; return nullptr; }
        owned.push_back(d);
        if (hipMemcpyAsync(d, user, sizeof(T) * n, hipMemcpyHostToDevice, s) != hipSuccess) bad = true;
        return d;
    }
    double* inout(double* user, size_t n, bool copyIn)
    {
        if (!user || fs_dev_ptr(user)) return user;
        double* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(double) * (n ? n : 1)) != hipSuccess) { bad = true; return nullptr; }
        owned.push_back(d);
        if (copyIn && hipMemcpyAsync(d, user, sizeof(double) * n, hipMemcpyHostToDevice, s) != hipSuccess) bad = true;
        return d;
    }
    int finish(double* user, double* dev, size_t n)
    {
        if (user && user != dev) LDU_CHECK_HIP(hipMemcpyAsync(user, dev, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        return 0;
    }
    int sync() { LDU_CHECK_HIP(hipStreamSynchronize(s)); return 0; }
};

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 ld3(const double* p, long i) { return D3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(double* p, long i, D3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
__device__ __forceinline__ double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // VectorI.H operator&
__device__ __forceinline__ double mag3(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }      // sqrt(magSqr)
__device__ __forceinline__ double fmax2(double a, double b) { return a > b ? a : b; }                   // Foam::max(scalar, scalar)

// ---------------------------------------------------------------- geometry

// surfaceInterpolation.C:289-305 (nonOrthDeltaCoeffs) and :346-352 (nonOrthCorrectionVectors), internal faces
__global__ void __launch_bounds__(FS_BLK)
fs_nonorth_kernel(int nF, const int* __restrict__ own, const int* __restrict__ nei, const double* __restrict__ Sf,
                  const double* __restrict__ magSf, const double* __restrict__ C, double* __restrict__ nod,
                  double* __restrict__ corr)
{
    for (long f = (long)blockIdx.x * FS_BLK + threadIdx.x; f < nF; f += (long)gridDim.x * FS_BLK)
    {
        const D3 cn = ld3(C, nei[f]), co = ld3(C, own[f]);
        const D3 delta{cn.x - co.x, cn.y - co.y, cn.z - co.z};
        const D3 sf = ld3(Sf, f);
        const double ms = magSf ? magSf[f] : mag3(sf) + kVSmall;
        const D3 ua{sf.x / ms, sf.y / ms, sf.z / ms};
        const double d = 1.0 / fmax2(dot3(ua, delta), 0.05 * mag3(delta));
        if (nod) nod[f] = d;
        if (corr) st3(corr, f, D3{ua.x - delta.x * d, ua.y - delta.y * d, ua.z - delta.z * d});
    }
}

// the patch faces: :307-313 and :362-390 (correction vectors only on coupled patches)
__global__ void __launch_bounds__(FS_BLK)
fs_nonorth_patch_kernel(int n, const double* __restrict__ Sf, const double* __restrict__ magSf,
                        const double* __restrict__ delta3, int coupled, double* __restrict__ nod,
                        double* __restrict__ corr)
{
    for (long f = (long)blockIdx.x * FS_BLK + threadIdx.x; f < n; f += (long)gridDim.x * FS_BLK)
    {
        const D3 sf = ld3(Sf, f), delta = ld3(delta3, f);
        const double ms = magSf ? magSf[f] : mag3(sf) + kVSmall;
        const D3 nf{sf.x / ms, sf.y / ms, sf.z / ms};
        const double d = 1.0 / fmax2(dot3(nf, delta), 0.05 * mag3(delta));
        if (nod) nod[f] = d;
        if (corr)
            st3(corr, f, coupled ? D3{nf.x - delta.x * d, nf.y - delta.y * d, nf.z - delta.z * d} : D3{0.0, 0.0, 0.0});
    }
}

// gaussLaplacianScheme.C:165-173: Sn = Sf/magSf; SfGamma = Sf & gamma; SfGammaSn = SfGamma & Sn;
// SfGammaCorr = SfGamma - SfGammaSn*Sn   (NG = 6: SymmTensorI.H operator&(Vector, SymmTensor); 9: TensorI.H)
template <int NG>
__global__ void __launch_bounds__(FS_BLK)
fs_tensorGamma_kernel(int n, const double* __restrict__ Sf, const double* __restrict__ magSf,
                      const double* __restrict__ gamma, double* __restrict__ sfGammaSn, double* __restrict__ sfGammaCorr)
{
    for (long f = (long)blockIdx.x * FS_BLK + threadIdx.x; f < n; f += (long)gridDim.x * FS_BLK)
    {
        const D3 sf = ld3(Sf, f);
        const double ms = magSf ? magSf[f] : mag3(sf) + kVSmall;
        const D3 sn{sf.x / ms, sf.y / ms, sf.z / ms};
        const double* g = gamma + (long)NG * f;
        D3 sg;
        if (NG == 6)
        {
            sg.x = sf.x * g[0] + sf.y * g[1] + sf.z * g[2];
            sg.y = sf.x * g[1] + sf.y * g[3] + sf.z * g[4];
            sg.z = sf.x * g[2] + sf.y * g[4] + sf.z * g[5];
        }
        else
        {
            sg.x = sf.x * g[0] + sf.y * g[3] + sf.z * g[6];
            sg.y = sf.x * g[1] + sf.y * g[4] + sf.z * g[7];
            sg.z = sf.x * g[2] + sf.y * g[5] + sf.z * g[8];
        }
        const double s = dot3(sg, sn);
        if (sfGammaSn) sfGammaSn[f] = s;
        if (sfGammaCorr) st3(sfGammaCorr, f, D3{sg.x - s * sn.x, sg.y - s * sn.y, sg.z - s * sn.z});
    }
}

// ---------------------------------------------------------------- face algebra

// Vector & Type per face: NC = 3 (Type vector -> scalar), 9 (tensor -> vector, TensorI.H operator&(Vector, Tensor))
template <int NC>
__device__ __forceinline__ void fs_dot(D3 v, const double* g, double* out)
{
    if (NC == 3) out[0] = v.x * g[0] + v.y * g[1] + v.z * g[2];
    else
    {
#pragma unroll
        for (int j = 0; j < 3; j++) out[j] = v.x * g[j] + v.y * g[3 + j] + v.z * g[6 + j];
    }
}

template <int NC>
__global__ void __launch_bounds__(FS_BLK)
fs_faceDot_kernel(int n, const double* __restrict__ vec, const double* __restrict__ field, double* __restrict__ out)
{
    for (long f = (long)blockIdx.x * FS_BLK + threadIdx.x; f < n; f += (long)gridDim.x * FS_BLK)
    {
        double g[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) g[q] = field[(long)NC * f + q];
        fs_dot<NC>(ld3(vec, f), g, out + (long)(NC / 3) * f);
    }
}

// vec & linear.interpolate(field): surfaceInterpolationScheme.C:293-296 (lambda*(vf[P] - vf[N]) + vf[N]) then the
// inner product.  vec = nonOrthCorrectionVectors (correctedSnGrad.C:44-66), Sf (gaussDivScheme.C:60-63) or
// SfGammaCorr (gaussLaplacianScheme.C:119-125)
template <int NC>
__global__ void __launch_bounds__(FS_BLK)
fs_interpolateDot_kernel(int nF, const int* __restrict__ own, const int* __restrict__ nei,
                         const double* __restrict__ vec, const double* __restrict__ w,
                         const double* __restrict__ field, double* __restrict__ out)
{
    for (long f = (long)blockIdx.x * FS_BLK + threadIdx.x; f < nF; f += (long)gridDim.x * FS_BLK)
    {
        const long P = own[f], N = nei[f];
        const double lam = w[f];
        double g[NC];
#pragma unroll
        for (int q = 0; q < NC; q++)
        {
            const double a = field[NC * P + q], b = field[NC * N + q];
            g[q] = lam * (a - b) + b;
        }
        fs_dot<NC>(ld3(vec, f), g, out + (long)(NC / 3) * f);
    }
}

// out = scale*field  |  out += scale*field   (surfaceScalarField * surfaceField<Type>)
__global__ void __launch_bounds__(FS_BLK)
fs_faceScale_kernel(long n, int nComp, const double* __restrict__ scale, const double* __restrict__ field, int accumulate,
                    double* __restrict__ out)
{
    for (long i = (long)blockIdx.x * FS_BLK + threadIdx.x; i < n * nComp; i += (long)gridDim.x * FS_BLK)
    {
        const double v = scale[i / nComp] * field[i];
        out[i] = accumulate ? out[i] + v : v;
    }
}

// snGradScheme.C:139-143 with nonOrthDeltaCoeffs, then += correction (:176-180)
__global__ void __launch_bounds__(FS_BLK)
fs_correctedSnGrad_kernel(int nF, int nComp, const int* __restrict__ own, const int* __restrict__ nei,
                          const double* __restrict__ delta, const double* __restrict__ vf,
                          const double* __restrict__ corr, double* __restrict__ ssf)
{
    for (long i = (long)blockIdx.x * FS_BLK + threadIdx.x; i < (long)nF * nComp; i += (long)gridDim.x * FS_BLK)
    {
        const long f = i / nComp;
        const int q = (int)(i - f * nComp);
        const double s = delta[f] * (vf[(long)nComp * nei[f] + q] - vf[(long)nComp * own[f] + q]);
        ssf[i] = corr ? s + corr[i] : s;
    }
}

// ---------------------------------------------------------------- patch faces

// surfaceInterpolationScheme.C:298-314
__global__ void __launch_bounds__(FS_BLK)
fs_interpolateBoundary_kernel(int nB, int nComp, const int* __restrict__ faceCells,
                              const unsigned char* __restrict__ coupled, const double* __restrict__ w,
                              const double* __restrict__ vf, const double* __restrict__ pnf,
                              const double* __restrict__ values, double* __restrict__ out)
{
    for (long i = (long)blockIdx.x * FS_BLK + threadIdx.x; i < (long)nB * nComp; i += (long)gridDim.x * FS_BLK)
    {
        const long f = i / nComp;
        const int q = (int)(i - f * nComp);
        if (coupled[f])
        {
            const double lam = w[f];
            out[i] = lam * vf[(long)nComp * faceCells[f] + q] + (1.0 - lam) * pnf[i];
        }
        else if (values) out[i] = values[i];
    }
}

// gaussGrad.C:144-170 on the ordinary patches: g = grad[faceCell] (zeroGradient patch field of the gradient);
// g += n*(snGrad - (n & g)).  NC = 1: scalar field (g vector), 3: vector field (g tensor, n*d the outer product)
template <int NC>
__global__ void __launch_bounds__(FS_BLK)
fs_gaussGradBoundary_kernel(int nB, const int* __restrict__ faceCells, const unsigned char* __restrict__ coupled,
                            const double* __restrict__ nf3, const double* __restrict__ grad,
                            const double* __restrict__ snGrad, double* __restrict__ out)
{
    for (long f = (long)blockIdx.x * FS_BLK + threadIdx.x; f < nB; f += (long)gridDim.x * FS_BLK)
    {
        if (coupled[f]) continue;
        const D3 n = ld3(nf3, f);
        double g[3 * NC];
#pragma unroll
        for (int q = 0; q < 3 * NC; q++) g[q] = grad[(long)3 * NC * faceCells[f] + q];
        double d[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) d[j] = snGrad[(long)NC * f + j] - (n.x * g[j] + n.y * g[NC + j] + n.z * g[2 * NC + j]);
        const double nn[3] = {n.x, n.y, n.z};
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < NC; j++) out[(long)3 * NC * f + NC * i + j] = g[NC * i + j] + nn[i] * d[j];
    }
}

// ---------------------------------------------------------------- cell gathers

// fvc::surfaceIntegrate with its patch faces (fvcSurfaceIntegrate.C:43-76); MODE 0: out = acc/V,
// MODE 1: source -= V*(acc/V)  (gaussLaplacianSchemes.C:74-88, gaussLaplacianScheme.C:187)
template <int NC, int MODE>
__global__ void __launch_bounds__(FS_BLK)
fs_surfaceIntegrate_kernel(int nCells, const int* __restrict__ cs, const int* __restrict__ cf,
                           const int* __restrict__ losortStart, const int* __restrict__ losort,
                           const int* __restrict__ ownerStart, const double* __restrict__ ssf,
                           const double* __restrict__ bssf, const double* __restrict__ V, double* __restrict__ out)
{
  for (long c = (long)blockIdx.x * FS_BLK + threadIdx.x; c < nCells; c += (long)gridDim.x * FS_BLK)
  {
    double acc[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) acc[q] = 0.0;
    // faces four at a time: index loads, then value loads, in flight together (the sums stay in face order)
    const int t1 = losortStart[c + 1];
    for (int t = losortStart[c]; t < t1; t += 4)
    {
        long f[4];
        double v[4][NC];
#pragma unroll
        for (int k = 0; k < 4; k++) f[k] = losort[t + k < t1 ? t + k : t1 - 1];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int q = 0; q < NC; q++) v[k][q] = ssf[NC * f[k] + q];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (t + k < t1)
            {
#pragma unroll
                for (int q = 0; q < NC; q++) acc[q] -= v[k][q];
            }
    }
    const long f1 = ownerStart[c + 1];
    for (long f0 = ownerStart[c]; f0 < f1; f0 += 4)
    {
        double v[4][NC];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const long f = f0 + k < f1 ? f0 + k : f1 - 1;
#pragma unroll
            for (int q = 0; q < NC; q++) v[k][q] = ssf[NC * f + q];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (f0 + k < f1)
            {
#pragma unroll
                for (int q = 0; q < NC; q++) acc[q] += v[k][q];
            }
    }
    if (cs)
        for (int j = cs[c]; j < cs[c + 1]; j++)
        {
            const long f = cf[j];
#pragma unroll
            for (int q = 0; q < NC; q++) acc[q] += bssf[NC * f + q];
        }
    const double v = V[c];
#pragma unroll
    for (int q = 0; q < NC; q++)
    {
        if (MODE == 0) out[NC * c + q] = acc[q] / v;
        else out[NC * c + q] -= v * (acc[q] / v);
    }
  }
}

// ---------------------------------------------------------------- C ABI
extern "C" {

int ldu_mesh_nonorth_factors(ldu_ctx* ctx, int32_t nCells, int32_t nInternalFaces, const int32_t* owner,
                             const int32_t* neighbour, const double* faceAreas, const double* magSf,
                             const double* cellCentres, double* nonOrthDeltaCoeffs, double* nonOrthCorrectionVectors)
{
    if (!ctx || nCells < 0 || nInternalFaces < 0) { ldu_set_error("ldu_mesh_nonorth_factors: bad sizes"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    FsBuf B(ctx->stream);
    const size_t nF = (size_t)nInternalFaces;
    const int* o = B.in(owner, nF);
    const int* n = B.in(neighbour, nF);
    const double* sf = B.in(faceAreas, 3 * nF);
    const double* ms = B.in(magSf, nF);
    const double* c = B.in(cellCentres, 3 * (size_t)nCells);
    double* d = B.inout(nonOrthDeltaCoeffs, nF, false);
    double* cv = B.inout(nonOrthCorrectionVectors, 3 * nF, false);
    if (B.failed()) return -1;
    if (nF)
        fs_nonorth_kernel<<<fs_grid(nInternalFaces), FS_BLK, 0, B.s>>>(nInternalFaces, o, n, sf, ms, c, d, cv);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(nonOrthDeltaCoeffs, d, nF) || B.finish(nonOrthCorrectionVectors, cv, 3 * nF)) return -1;
    return B.sync();
}

int ldu_mesh_patch_nonorth_factors(ldu_ctx* ctx, int32_t nPatchFaces, const double* patchSf, const double* patchMagSf,
                                   const double* patchDelta, int32_t coupled, double* nonOrthDeltaCoeffs,
                                   double* nonOrthCorrectionVectors)
{
    if (!ctx || nPatchFaces < 0) { ldu_set_error("ldu_mesh_patch_nonorth_factors: bad sizes"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    FsBuf B(ctx->stream);
    const size_t n = (size_t)nPatchFaces;
    const double* sf = B.in(patchSf, 3 * n);
    const double* ms = B.in(patchMagSf, n);
    const double* dl = B.in(patchDelta, 3 * n);
    double* d = B.inout(nonOrthDeltaCoeffs, n, false);
    double* cv = B.inout(nonOrthCorrectionVectors, 3 * n, false);
    if (B.failed()) return -1;
    if (n) fs_nonorth_patch_kernel<<<fs_grid(nPatchFaces), FS_BLK, 0, B.s>>>(nPatchFaces, sf, ms, dl, coupled, d, cv);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(nonOrthDeltaCoeffs, d, n) || B.finish(nonOrthCorrectionVectors, cv, 3 * n)) return -1;
    return B.sync();
}

int ldu_fv_tensorGammaFactors(ldu_ctx* ctx, int32_t nFaces, int32_t nGammaCmpt, const double* Sf, const double* magSf,
                              const double* gamma, double* SfGammaSn, double* SfGammaCorr)
{
    if (!ctx || nFaces < 0 || (nGammaCmpt != 6 && nGammaCmpt != 9))
    {
        ldu_set_error("ldu_fv_tensorGammaFactors: nGammaCmpt must be 6 (symmTensor) or 9 (tensor)");
        return -14;
    }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    FsBuf B(ctx->stream);
    const size_t n = (size_t)nFaces;
    const double* sf = B.in(Sf, 3 * n);
    const double* ms = B.in(magSf, n);
    const double* g = B.in(gamma, (size_t)nGammaCmpt * n);
    double* a = B.inout(SfGammaSn, n, false);
    double* c = B.inout(SfGammaCorr, 3 * n, false);
    if (B.failed()) return -1;
    if (n)
    {
        if (nGammaCmpt == 6) fs_tensorGamma_kernel<6><<<fs_grid(nFaces), FS_BLK, 0, B.s>>>(nFaces, sf, ms, g, a, c);
        else fs_tensorGamma_kernel<9><<<fs_grid(nFaces), FS_BLK, 0, B.s>>>(nFaces, sf, ms, g, a, c);
    }
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(SfGammaSn, a, n) || B.finish(SfGammaCorr, c, 3 * n)) return -1;
    return B.sync();
}

int ldu_fv_faceDot(ldu_ctx* ctx, int32_t nFaces, int32_t nComp, const double* vec, const double* field, double* out)
{
    if (!ctx || nFaces < 0 || (nComp != 3 && nComp != 9)) { ldu_set_error("ldu_fv_faceDot: nComp must be 3 or 9"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    FsBuf B(ctx->stream);
    const size_t n = (size_t)nFaces, nOut = n * (size_t)(nComp / 3);
    const double* v = B.in(vec, 3 * n);
    const double* f = B.in(field, (size_t)nComp * n);
    double* o = B.inout(out, nOut, false);
    if (B.failed()) return -1;
    if (n)
    {
        if (nComp == 3) fs_faceDot_kernel<3><<<fs_grid(nFaces), FS_BLK, 0, B.s>>>(nFaces, v, f, o);
        else fs_faceDot_kernel<9><<<fs_grid(nFaces), FS_BLK, 0, B.s>>>(nFaces, v, f, o);
    }
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(out, o, nOut)) return -1;
    return B.sync();
}

int ldu_fv_faceScale(ldu_ctx* ctx, int32_t nFaces, int32_t nComp, const double* scale, const double* field,
                     int32_t accumulate, double* out)
{
    if (!ctx || nFaces < 0 || nComp < 1) { ldu_set_error("ldu_fv_faceScale: bad sizes"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    FsBuf B(ctx->stream);
    const size_t n = (size_t)nFaces * nComp;
    const double* s = B.in(scale, (size_t)nFaces);
    const double* f = B.in(field, n);
    double* o = B.inout(out, n, accumulate != 0);
    if (B.failed()) return -1;
    if (n) fs_faceScale_kernel<<<fs_grid((long)n), FS_BLK, 0, B.s>>>(nFaces, nComp, s, f, accumulate, o);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(out, o, n)) return -1;
    return B.sync();
}

int ldu_fv_interpolateDot(ldu_addr* a, int32_t nComp, const double* vec, const double* weights, const double* field,
                          double* out)
{
    if (nComp != 3 && nComp != 9) { ldu_set_error("ldu_fv_interpolateDot: nComp must be 3 or 9"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(a->ctx->device));
    FsBuf B(a->ctx->stream);
    const size_t nF = (size_t)a->nFaces, nOut = nF * (size_t)(nComp / 3);
    const double* v = B.in(vec, 3 * nF);
    const double* w = B.in(weights, nF);
    const double* f = B.in(field, (size_t)nComp * a->nCells);
    double* o = B.inout(out, nOut, false);
    if (B.failed()) return -1;
    if (nF)
    {
        if (nComp == 3) fs_interpolateDot_kernel<3><<<fs_grid(a->nFaces), FS_BLK, 0, B.s>>>(a->nFaces, a->d_l, a->d_u, v, w, f, o);
        else fs_interpolateDot_kernel<9><<<fs_grid(a->nFaces), FS_BLK, 0, B.s>>>(a->nFaces, a->d_l, a->d_u, v, w, f, o);
    }
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(out, o, nOut)) return -1;
    return B.sync();
}

int ldu_fvc_correctedSnGrad(ldu_addr* a, int32_t nComp, const double* nonOrthDeltaCoeffs, const double* vf,
                            const double* correction, double* ssf)
{
    if (nComp < 1) { ldu_set_error("ldu_fvc_correctedSnGrad: bad nComp"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(a->ctx->device));
    FsBuf B(a->ctx->stream);
    const size_t n = (size_t)a->nFaces * nComp;
    const double* d = B.in(nonOrthDeltaCoeffs, (size_t)a->nFaces);
    const double* v = B.in(vf, (size_t)a->nCells * nComp);
    const double* c = B.in(correction, n);
    double* o = B.inout(ssf, n, false);
    if (B.failed()) return -1;
    if (n) fs_correctedSnGrad_kernel<<<fs_grid((long)n), FS_BLK, 0, B.s>>>(a->nFaces, nComp, a->d_l, a->d_u, d, v, c, o);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(ssf, o, n)) return -1;
    return B.sync();
}

int ldu_fv_interpolateBoundary(ldu_fv_boundary* b, int32_t nComp, const double* patchWeights, const double* vf,
                               const double* patchNeighbourField, const double* patchValues, double* out)
{
    if (!b || nComp < 1) { ldu_set_error("ldu_fv_interpolateBoundary: bad arguments"); return -14; }
    ldu_addr* a = b->a;
    LDU_CHECK_HIP(hipSetDevice(a->ctx->device));
    FsBuf B(a->ctx->stream);
    const size_t n = (size_t)b->nFacesTotal * nComp;
    const double* w = B.in(patchWeights, (size_t)b->nFacesTotal);
    const double* v = B.in(vf, (size_t)a->nCells * nComp);
    const double* pn = B.in(patchNeighbourField, n);
    const double* pv = B.in(patchValues, n);
    double* o = B.inout(out, n, patchValues == nullptr);
    if (B.failed()) return -1;
    if (n)
        fs_interpolateBoundary_kernel<<<fs_grid((long)n), FS_BLK, 0, B.s>>>(b->nFacesTotal, nComp, b->d_faceCells, b->d_coupled,
                                                                          w, v, pn, pv, o);
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(out, o, n)) return -1;
    return B.sync();
}

int ldu_fvc_gaussGradBoundary(ldu_fv_boundary* b, int32_t nComp, const double* patchNf, const double* grad,
                              const double* patchSnGrad, double* boundaryGrad)
{
    if (!b || (nComp != 1 && nComp != 3)) { ldu_set_error("ldu_fvc_gaussGradBoundary: nComp must be 1 or 3"); return -14; }
    ldu_addr* a = b->a;
    LDU_CHECK_HIP(hipSetDevice(a->ctx->device));
    FsBuf B(a->ctx->stream);
    const size_t nB = (size_t)b->nFacesTotal;
    const double* nf = B.in(patchNf, 3 * nB);
    const double* g = B.in(grad, 3 * (size_t)nComp * a->nCells);
    const double* sg = B.in(patchSnGrad, (size_t)nComp * nB);
    double* o = B.inout(boundaryGrad, 3 * (size_t)nComp * nB, true);   // coupled faces keep what the caller put there
    if (B.failed()) return -1;
    if (nB)
    {
        if (nComp == 1)
            fs_gaussGradBoundary_kernel<1><<<fs_grid(b->nFacesTotal), FS_BLK, 0, B.s>>>(b->nFacesTotal, b->d_faceCells, b->d_coupled, nf, g, sg, o);
        else
            fs_gaussGradBoundary_kernel<3><<<fs_grid(b->nFacesTotal), FS_BLK, 0, B.s>>>(b->nFacesTotal, b->d_faceCells, b->d_coupled, nf, g, sg, o);
    }
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(boundaryGrad, o, 3 * (size_t)nComp * nB)) return -1;
    return B.sync();
}

static int fs_integrate(ldu_addr* a, ldu_fv_boundary* b, int nComp, const double* ssf, const double* boundarySsf,
                        const double* V, double* out, int mode, const char* who)
{
    if (nComp != 1 && nComp != 3) { ldu_set_error(std::string(who) + ": nComp must be 1 or 3"); return -14; }
    if (b && b->a != a) { ldu_set_error(std::string(who) + ": boundary belongs to another addressing"); return -2; }
    LDU_CHECK_HIP(hipSetDevice(a->ctx->device));
    FsBuf B(a->ctx->stream);
    const size_t nB = b ? (size_t)b->nFacesTotal : 0;
    const double* f = B.in(ssf, (size_t)nComp * a->nFaces);
    const double* bf = B.in(boundarySsf, (size_t)nComp * nB);
    const double* v = B.in(V, (size_t)a->nCells);
    double* o = B.inout(out, (size_t)nComp * a->nCells, mode == 1);
    if (B.failed()) return -1;
    const int* cs = (b && boundarySsf) ? b->d_cellStart : nullptr;
    const int* cf = (b && boundarySsf) ? b->d_cellFace : nullptr;
    const int grid = fs_grid(a->nCells);
    if (a->nCells)
    {
        if (nComp == 1 && mode == 0)
            fs_surfaceIntegrate_kernel<1, 0><<<grid, FS_BLK, 0, B.s>>>(a->nCells, cs, cf, a->d_losortStart, a->d_losort, a->d_ownerStart, f, bf, v, o);
        else if (nComp == 1)
            fs_surfaceIntegrate_kernel<1, 1><<<grid, FS_BLK, 0, B.s>>>(a->nCells, cs, cf, a->d_losortStart, a->d_losort, a->d_ownerStart, f, bf, v, o);
        else if (mode == 0)
            fs_surfaceIntegrate_kernel<3, 0><<<grid, FS_BLK, 0, B.s>>>(a->nCells, cs, cf, a->d_losortStart, a->d_losort, a->d_ownerStart, f, bf, v, o);
        else
            fs_surfaceIntegrate_kernel<3, 1><<<grid, FS_BLK, 0, B.s>>>(a->nCells, cs, cf, a->d_losortStart, a->d_losort, a->d_ownerStart, f, bf, v, o);
    }
    LDU_CHECK_HIP(hipGetLastError());
    if (B.finish(out, o, (size_t)nComp * a->nCells)) return -1;
    return B.sync();
}

int ldu_fvc_surfaceIntegrateFull(ldu_addr* a, ldu_fv_boundary* b, int32_t nComp, const double* ssf,
                                 const double* boundarySsf, const double* V, double* out)
{
    return fs_integrate(a, b, nComp, ssf, boundarySsf, V, out, 0, "ldu_fvc_surfaceIntegrateFull");
}

int ldu_fvm_sourceMinusVDiv(ldu_addr* a, ldu_fv_boundary* b, int32_t nComp, const double* faceFluxCorrection,
                            const double* boundaryFaceFluxCorrection, const double* V, double* source)
{
    return fs_integrate(a, b, nComp, faceFluxCorrection, boundaryFaceFluxCorrection, V, source, 1, "ldu_fvm_sourceMinusVDiv");
}

}  // extern "C"
