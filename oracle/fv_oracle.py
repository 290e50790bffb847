"""TEST INFRASTRUCTURE - numpy restatement of the finite-volume face stencils that feed the
lduMatrix, internal faces only.  Each function follows the cited loop of
/root/reference/src/finiteVolume/... literally: np.add.at / np.subtract.at apply the updates one
face at a time in ascending face order, i.e. in the reference's accumulation order.

PINNED against the reference's own libfiniteVolume: oracle/build_ref_fv.sh compiles its units from
/root/reference, oracle/fv_driver.C runs them on a perturbed graded hex box, the vectors are committed
as tests/golden/fv_*.npz (tests/golden/make_fv_golden.py) and tests/test_fv_oracle_golden.py requires
every function below to reproduce them bit for bit.
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


def upwind_weights(phi):
    """upwind::limiter / weights, finiteVolume/interpolation/surfaceInterpolation/limitedSchemes/upwind/
    upwind.H: weights = pos(faceFlux)  (pos(0) = 1)"""
    return (phi >= 0).astype(np.float64)


def face_area_pair_weights(Sf, magSf):
    """faceAreaPairGAMGAgglomeration.C:48-73: mag(cmptMultiply(Sf/sqrt(magSf), vector(1, 1.01, 1.02)))"""
    t = Sf / np.sqrt(magSf)[:, None]
    t = t * np.array([1.0, 1.01, 1.02])
    return np.sqrt(t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1] + t[:, 2] * t[:, 2])
