"""TEST INFRASTRUCTURE - numpy restatement of the finite-volume face stencils that feed the
lduMatrix, internal faces only.  Each function follows the cited loop of
/root/reference/src/finiteVolume/... literally: np.add.at / np.subtract.at apply the updates one
face at a time in ascending face order, i.e. in the reference's accumulation order.

PARITY UNPINNED: libfiniteVolume is not built in this container (needs ~1000 units, SURVEY.md 8c
Tier 2), so these restatements have not been executed against the reference.  negSumDiag is the
exception: it lives in libOpenFOAM and is pinned through lduMatrix (test_oracle_vs_ref.py).
"""
import numpy as np


def interpolate(l, u, lambdas, vf):
    """surfaceInterpolationScheme::interpolate, interpolation/surfaceInterpolation/
    surfaceInterpolationScheme/surfaceInterpolationScheme.C:293-296:
    sf[f] = lambda[f]*(vf[P[f]] - vf[N[f]]) + vf[N[f]]"""
    lam = lambdas if vf.ndim == 1 else lambdas[:, None]
    return lam * (vf[l] - vf[u]) + vf[u]


def surface_integrate(l, u, ssf, V):
    """fvc::surfaceIntegrate, finiteVolume/fvc/fvcSurfaceIntegrate.C:56-60,75:
    ivf[own[f]] += ssf[f]; ivf[nei[f]] -= ssf[f]; ivf /= V"""
    out = np.zeros((V.size,) + ssf.shape[1:])
    for f in range(l.size):          # literal face loop: interleaves += and -= per face
        out[l[f]] += ssf[f]
        out[u[f]] -= ssf[f]
    return out / (V if ssf.ndim == 1 else V[:, None])


def gauss_grad(l, u, Sf, ssf, V):
    """fv::gaussGrad::gradf, finiteVolume/gradSchemes/gaussGrad/gaussGrad.C:82-88,105:
    Sfssf = Sf[f]*ssf[f]; g[own] += Sfssf; g[nei] -= Sfssf; g /= V"""
    return surface_integrate(l, u, Sf * ssf[:, None], V)


def sn_grad(l, u, delta, vf):
    """snGradScheme::snGrad, finiteVolume/snGradSchemes/snGradScheme/snGradScheme.C:139-143"""
    return delta * (vf[u] - vf[l])


def neg_sum_diag(nC, l, u, lower, upper):
    """lduMatrix::negSumDiag, OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixOperations.C:50-64"""
    d = np.zeros(nC)
    for f in range(l.size):
        d[l[f]] -= lower[f]
        d[u[f]] -= upper[f]
    return d


def fvm_laplacian(nC, l, u, delta, gammaMagSf):
    """gaussLaplacianScheme::fvmLaplacianUncorrected, finiteVolume/laplacianSchemes/
    gaussLaplacianScheme/gaussLaplacianScheme.C:63-64: upper = deltaCoeffs*gammaMagSf; negSumDiag"""
    upper = delta * gammaMagSf
    return neg_sum_diag(nC, l, u, upper, upper), upper


def fvm_div(nC, l, u, w, phi):
    """gaussConvectionScheme::fvmDiv, finiteVolume/convectionSchemes/gaussConvectionScheme/
    gaussConvectionScheme.C:87-89: lower = -w*phi; upper = lower + phi; negSumDiag"""
    lower = -w * phi
    upper = lower + phi
    return neg_sum_diag(nC, l, u, lower, upper), upper, lower
