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


# ---------------------------------------------------------------- fvMatrix glue
# scalar matrices; patches = list of dicts {faceCells, internalCoeffs, boundaryCoeffs, coupled, pnf}

def add_boundary_diag(diag, patches):
    """fvMatrix::addBoundaryDiag, fvMatrices/fvMatrix/fvMatrix.C:116-131 (cmpt 0 of a scalar)"""
    d = diag.copy()
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            d[c] += p["internalCoeffs"][i]
    return d


def add_boundary_source(source, patches, couples=True):
    """fvMatrix::addBoundarySource, fvMatrix.C:150-178"""
    s = source.copy()
    for p in patches:
        if not p["coupled"]:
            for i, c in enumerate(p["faceCells"]):
                s[c] += p["boundaryCoeffs"][i]
        elif couples:
            for i, c in enumerate(p["faceCells"]):
                s[c] += p["boundaryCoeffs"][i] * p["pnf"][i]
    return s


def sum_mag_off_diag(nC, l, u, lower, upper):
    """lduMatrix::sumMagOffDiag, OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixOperations.C:67-83"""
    s = np.zeros(nC)
    for f in range(l.size):
        s[u[f]] += abs(lower[f])
        s[l[f]] += abs(upper[f])
    return s


def relax(alpha, diag, source, l, u, upper, lower, psi, patches):
    """fvMatrix<scalar>::relax, fvMatrix.C:525-655"""
    if alpha <= 0:
        return diag.copy(), source.copy()
    D = diag.copy()
    D0 = diag.copy()
    sumOff = sum_mag_off_diag(D.size, l, u, lower, upper)
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            if p["coupled"]:
                D[c] += p["internalCoeffs"][i]
                sumOff[c] += abs(p["boundaryCoeffs"][i])
            else:
                D[c] += abs(p["internalCoeffs"][i])      # cmptMax(cmptMag(.))
    D = np.maximum(np.abs(D), sumOff)
    D = D / alpha
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            D[c] -= p["internalCoeffs"][i]               # component 0 (coupled) / cmptMin (non-coupled)
    return D, source + (D - D0) * psi


def set_reference(celli, value, diag, source):
    """fvMatrix::setReference, fvMatrix.C:509-521"""
    d, s = diag.copy(), source.copy()
    if celli >= 0:
        s[celli] += d[celli] * value
        d[celli] += d[celli]
    return d, s


def fvm_A(diag, patches, V):
    """fvMatrix::A, fvMatrix.C:722-746: D()/V with D() = diag + cmptAv(internalCoeffs) (:689-694)"""
    return add_boundary_diag(diag, patches) / V


def ldu_H(nC, l, u, lower, upper, psi):
    """lduMatrix::H, lduMatrixTemplates.C:32-69"""
    h = np.zeros(nC)
    for f in range(l.size):
        h[u[f]] -= lower[f] * psi[l[f]]
        h[l[f]] -= upper[f] * psi[u[f]]
    return h


def fvm_H(diag, source, l, u, upper, lower, psi, patches, V):
    """fvMatrix<scalar>::H - the scalar SPECIALISATION, fvMatrices/fvScalarMatrix/fvScalarMatrix.C:209-237:
    (lduMatrix::H(psi) + source), addBoundarySource, /V.  (The generic fvMatrix<Type>::H, fvMatrix.C:751-813,
    also adds (cmptAv(internalCoeffs) - internalCoeffs.component(cmpt))*psi, which is not exactly zero in
    floating point when a cell has several boundary faces.)"""
    h = ldu_H(diag.size, l, u, lower, upper, psi) + source
    h = add_boundary_source(h, patches, couples=True)
    return h / V


def fvm_flux(l, u, upper, lower, psi, patches):
    """fvMatrix::flux, fvMatrix.C:865-943; internal part = lduMatrix::faceH (lduMatrixTemplates.C:92-97)"""
    fi = upper * psi[u] - lower * psi[l]
    fb = []
    for p in patches:
        ic = p["internalCoeffs"] * psi[p["faceCells"]]
        nc = p["boundaryCoeffs"] * p["pnf"] if p["coupled"] else p["boundaryCoeffs"]
        fb.append(ic - nc)
    return fi, fb


# ---- vector (3-component) matrices: coefficient arrays [n][3]
def cmpt_av(v):
    """VectorSpaceI.H:400-418: ((v0 + v1) + v2)/3"""
    return ((v[..., 0] + v[..., 1]) + v[..., 2]) / 3


def add_boundary_diag_cmpt(diag, patches, cmpt):
    """fvMatrix::addBoundaryDiag(diag, cmpt), fvMatrix.C:116-131"""
    d = diag.copy()
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            d[c] += p["internalCoeffs"][i, cmpt]
    return d


def add_boundary_source_v(source, patches, couples=True):
    """fvMatrix<vector>::addBoundarySource, fvMatrix.C:150-178 (cmptMultiply on coupled patches)"""
    s = source.copy()
    for p in patches:
        if not p["coupled"]:
            for i, c in enumerate(p["faceCells"]):
                s[c] += p["boundaryCoeffs"][i]
        elif couples:
            for i, c in enumerate(p["faceCells"]):
                s[c] += p["boundaryCoeffs"][i] * p["pnf"][i]
    return s


def relax_v(alpha, diag, source, l, u, upper, lower, psi, patches):
    """fvMatrix<vector>::relax, fvMatrix.C:525-655"""
    D = diag.copy()
    D0 = diag.copy()
    sumOff = sum_mag_off_diag(D.size, l, u, lower, upper)
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            ic = p["internalCoeffs"][i]
            if p["coupled"]:
                D[c] += ic[0]
                sumOff[c] += abs(p["boundaryCoeffs"][i][0])
            else:
                D[c] += max(max(abs(ic[0]), abs(ic[1])), abs(ic[2]))
    D = np.maximum(np.abs(D), sumOff)
    D = D / alpha
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            ic = p["internalCoeffs"][i]
            D[c] -= ic[0] if p["coupled"] else min(min(ic[0], ic[1]), ic[2])
    return D, source + ((D - D0)[:, None] * psi)


def fvm_A_v(diag, patches, V):
    """fvMatrix::A with D() = diag + cmptAv(internalCoeffs), fvMatrix.C:689-694, 722-746"""
    d = diag.copy()
    for p in patches:
        av = cmpt_av(p["internalCoeffs"])
        for i, c in enumerate(p["faceCells"]):
            d[c] += av[i]
    return d / V


def fvm_H_v(diag, source, l, u, upper, lower, psi, patches, V):
    """fvMatrix<vector>::H, the generic template fvMatrix.C:751-813"""
    nC = diag.size
    h = np.zeros((nC, 3))
    for k in range(3):
        bd = add_boundary_diag_cmpt(np.zeros(nC), patches, k)
        bd = -bd
        for p in patches:
            av = cmpt_av(p["internalCoeffs"])
            for i, c in enumerate(p["faceCells"]):
                bd[c] += av[i]
        h[:, k] = bd * psi[:, k]
    hl = np.zeros((nC, 3))
    for f in range(l.size):
        hl[u[f]] -= lower[f] * psi[l[f]]
        hl[l[f]] -= upper[f] * psi[u[f]]
    h = h + (hl + source)
    h = add_boundary_source_v(h, patches, couples=True)
    return h / V[:, None]


# ---------------------------------------------------------------- higher-order schemes
def linear_upwind_correction(l, u, phi, C, Cf, grad):
    """linearUpwind<scalar>::correction, interpolation/surfaceInterpolation/schemes/linearUpwind/
    linearUpwind.C:87-91 (internal faces)"""
    c = np.where(phi > 0, l, u)
    d = Cf - C[c]
    g = grad[c]
    return d[:, 0] * g[:, 0] + d[:, 1] * g[:, 1] + d[:, 2] * g[:, 2]


def cell_limited_grad(k, l, u, vsf, C, Cf, grad, patches):
    """cellLimitedGrad<scalar>::calcGrad, gradSchemes/limitedGradSchemes/cellLimitedGrad/
    cellLimitedGrads.C:46-196 + limitFace cellLimitedGrad.H:136-152.  patches: dicts faceCells, value, Cf"""
    VSMALL = 1.0e-300
    g = grad.copy()
    mx, mn = vsf.copy(), vsf.copy()
    for f in range(l.size):
        o, n = l[f], u[f]
        mx[o] = max(mx[o], vsf[n]); mn[o] = min(mn[o], vsf[n])
        mx[n] = max(mx[n], vsf[o]); mn[n] = min(mn[n], vsf[o])
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            mx[c] = max(mx[c], p["value"][i]); mn[c] = min(mn[c], p["value"][i])
    mx = mx - vsf
    mn = mn - vsf
    if k < 1.0:
        mm = (1.0 / k - 1.0) * (mx - mn)
        mx = mx + mm
        mn = mn - mm
    lim = np.ones(vsf.size)

    def limit(c, ex):
        if ex > mx[c] + VSMALL:
            lim[c] = min(lim[c], mx[c] / ex)
        elif ex < mn[c] - VSMALL:
            lim[c] = min(lim[c], mn[c] / ex)

    def dot(a, b):
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

    for f in range(l.size):
        o, n = l[f], u[f]
        limit(o, dot(Cf[f] - C[o], g[o]))
        limit(n, dot(Cf[f] - C[n], g[n]))
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            limit(c, dot(p["Cf"][i] - C[c], g[c]))
    return g * lim[:, None]


def linear_upwind_v_correction(l, u, phi, w, vf, C, Cf, gradT):
    """linearUpwindV<vector>::correction, interpolation/surfaceInterpolation/schemes/linearUpwind/linearUpwindV.C:
    87-140 (internal faces).  gradT: [nCells, 9] tensors (xx xy xz yx ...), vector & tensor = TensorI.H."""
    VSMALL = 1.0e-300
    out = np.zeros((l.size, 3))
    for f in range(l.size):
        o, n = l[f], u[f]
        if phi[f] > 0.0:
            maxCorr = (1.0 - w[f]) * (vf[n] - vf[o])
            d, g = Cf[f] - C[o], gradT[o]
        else:
            maxCorr = w[f] * (vf[o] - vf[n])
            d, g = Cf[f] - C[n], gradT[n]
        s = np.array([d[0] * g[0] + d[1] * g[3] + d[2] * g[6],
                      d[0] * g[1] + d[1] * g[4] + d[2] * g[7],
                      d[0] * g[2] + d[1] * g[5] + d[2] * g[8]])
        sfCorrs = s[0] * s[0] + s[1] * s[1] + s[2] * s[2]
        maxCorrs = s[0] * maxCorr[0] + s[1] * maxCorr[1] + s[2] * maxCorr[2]
        if sfCorrs > 0:
            if maxCorrs < 0:
                s = np.zeros(3)
            elif sfCorrs > maxCorrs:
                s = s * (maxCorrs / (sfCorrs + VSMALL))
        elif sfCorrs < 0:
            if maxCorrs > 0:
                s = np.zeros(3)
            elif sfCorrs < maxCorrs:
                s = s * (maxCorrs / (sfCorrs - VSMALL))
        out[f] = s
    return out


def cell_limited_grad_v(k, l, u, vsf, C, Cf, gradT, patches):
    """cellLimitedGrad<vector>::calcGrad, cellLimitedGrads.C:200-360 (component-wise limitFace,
    cellLimitedGrad.H:155-174; the limiter of component j scales column j of the gradient tensor).
    patches: dicts faceCells, value [n,3], Cf"""
    VSMALL = 1.0e-300
    g = gradT.copy()
    mx, mn = vsf.copy(), vsf.copy()
    for f in range(l.size):
        o, n = l[f], u[f]
        mx[o] = np.maximum(mx[o], vsf[n]); mn[o] = np.minimum(mn[o], vsf[n])
        mx[n] = np.maximum(mx[n], vsf[o]); mn[n] = np.minimum(mn[n], vsf[o])
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            mx[c] = np.maximum(mx[c], p["value"][i]); mn[c] = np.minimum(mn[c], p["value"][i])
    mx = mx - vsf
    mn = mn - vsf
    if k < 1.0:
        mm = (1.0 / k - 1.0) * (mx - mn)
        mx = mx + mm
        mn = mn - mm
    lim = np.ones_like(vsf)

    def limit(c, d):
        t = g[c]
        ex = (d[0] * t[0] + d[1] * t[3] + d[2] * t[6], d[0] * t[1] + d[1] * t[4] + d[2] * t[7],
              d[0] * t[2] + d[1] * t[5] + d[2] * t[8])
        for j in range(3):
            if ex[j] > mx[c, j] + VSMALL:
                lim[c, j] = min(lim[c, j], mx[c, j] / ex[j])
            elif ex[j] < mn[c, j] - VSMALL:
                lim[c, j] = min(lim[c, j], mn[c, j] / ex[j])

    for f in range(l.size):
        limit(l[f], Cf[f] - C[l[f]])
        limit(u[f], Cf[f] - C[u[f]])
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            limit(c, p["Cf"][i] - C[c])
    out = g.copy()
    for j in range(3):
        out[:, j] = lim[:, j] * g[:, j]
        out[:, 3 + j] = lim[:, j] * g[:, 3 + j]
        out[:, 6 + j] = lim[:, j] * g[:, 6 + j]
    return out


def bounded_sp(diag, l, u, phi, patches, V):
    """the `bounded` convection wrapper's implicit term, boundedConvectionScheme.C:60-77:
    fvmDiv - fvm::Sp(fvc::surfaceIntegrate(phi), vf)  ->  diag -= V * (surfaceIntegrate(phi)) with
    fvcSurfaceIntegrate.C:43-76 (own += / nei -= in face order, then the patch fluxes, then / V).
    patches: dicts faceCells, phi"""
    ivf = np.zeros(diag.size)
    for f in range(l.size):
        ivf[l[f]] += phi[f]
        ivf[u[f]] -= phi[f]
    for p in patches:
        for i, c in enumerate(p["faceCells"]):
            ivf[c] += p["phi"][i]
    ivf = ivf / V
    return diag - V * ivf


# ---------------------------------------------------------------- non-orthogonal correction (a36, a37), gaussDiv (a39)
# and the patch halves of interpolate (a35) / gaussGrad (a34).  Vectors are [n,3]; tensors [n,9] in the reference's
# component order xx xy xz yx yy yz zx zy zz; symmTensors [n,6] xx xy xz yy yz zz.

def _dot(a, b):
    """Vector & Vector (VectorI.H): a.x*b.x + a.y*b.y + a.z*b.z, left to right"""
    return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1] + a[:, 2] * b[:, 2]


def _mag(a):
    """mag(Vector) = sqrt(magSqr), magSqr = x*x + y*y + z*z"""
    return np.sqrt(a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1] + a[:, 2] * a[:, 2])


def vec_dot_field(v, g):
    """Vector & Type per face: Type = vector ([n,3] -> [n]) or tensor ([n,9] -> [n,3], TensorI.H
    operator&(Vector, Tensor): (v.x*t.xj + v.y*t.yj + v.z*t.zj))"""
    if g.shape[1] == 3:
        return _dot(v, g)
    return np.stack([v[:, 0] * g[:, j] + v[:, 1] * g[:, 3 + j] + v[:, 2] * g[:, 6 + j] for j in range(3)], axis=1)


def nonorth_factors(l, u, C, Sf, magSf):
    """surfaceInterpolation::makeNonOrthDeltaCoeffs / makeNonOrthCorrectionVectors, internal faces,
    interpolation/surfaceInterpolation/surfaceInterpolation/surfaceInterpolation.C:289-305, :346-352:
    delta = C[nei]-C[own]; unitArea = Sf/magSf; nonOrthDeltaCoeffs = 1/max(unitArea & delta, 0.05*mag(delta));
    corrVecs = unitArea - delta*nonOrthDeltaCoeffs"""
    delta = C[u] - C[l]
    unitArea = Sf / magSf[:, None]
    nod = 1.0 / np.maximum(_dot(unitArea, delta), 0.05 * _mag(delta))
    return nod, unitArea - delta * nod[:, None]


def nonorth_factors_patch(Sf, magSf, delta, coupled):
    """the patch faces, surfaceInterpolation.C:307-313 (1/max(nf & delta, 0.05*mag(delta)), nf = Sf/magSf:
    fvPatch.C nf()) and :362-390 (correction vectors: zero on ordinary patches, unitArea - delta*nonOrthDeltaCoeffs on
    coupled ones)"""
    nf = Sf / magSf[:, None]
    nod = 1.0 / np.maximum(_dot(nf, delta), 0.05 * _mag(delta))
    corr = nf - delta * nod[:, None] if coupled else np.zeros_like(Sf)
    return nod, corr


def interpolate_patch(w, vf, faceCells, pnf, values, coupled):
    """surfaceInterpolationScheme::interpolate on a patch, surfaceInterpolationScheme.C:298-314: coupled ->
    pLambda*patchInternalField + (1 - pLambda)*patchNeighbourField, otherwise the patch values"""
    if not coupled:
        return values.copy()
    lam = w if vf.ndim == 1 else w[:, None]
    return lam * vf[faceCells] + (1.0 - lam) * pnf


def gauss_grad_boundary(nf, gradInternalAtFaceCells, snGrad):
    """gaussGrad::correctBoundaryConditions, finiteVolume/gradSchemes/gaussGrad/gaussGrad.C:144-170, on an ordinary
    (not coupled) patch whose gradient patch field is zeroGradient (= the face cell's gradient):
    g += n*(snGrad - (n & g)); scalar field: g [n,3], snGrad [n]; vector field: g [n,9], snGrad [n,3]"""
    g = gradInternalAtFaceCells
    if g.shape[1] == 3:
        d = snGrad - _dot(nf, g)
        return g + nf * d[:, None]
    d = snGrad - vec_dot_field(nf, g)            # vector
    outer = np.stack([nf[:, i] * d[:, j] for i in range(3) for j in range(3)], axis=1)   # Vector * Vector (outer)
    return g + outer


def sn_grad_correction(l, u, w, corrVecs, grad):
    """correctedSnGrad<Type>::fullGradCorrection, snGradSchemes/correctedSnGrad/correctedSnGrad.C:44-66 (and the
    scalar / vector specialisations of correction(), correctedSnGrads.C:44-62): nonOrthCorrectionVectors &
    linear.interpolate(grad(vf)), internal faces"""
    return vec_dot_field(corrVecs, interpolate(l, u, w, grad))


def corrected_sn_grad(l, u, nonOrthDelta, vf, corr):
    """snGradScheme::snGrad(vf), snGradScheme.C:168-186: snGrad(vf, deltaCoeffs(vf)) += correction(vf) with
    deltaCoeffs = nonOrthDeltaCoeffs (correctedSnGrad.H:98-104)"""
    d = nonOrthDelta if vf.ndim == 1 else nonOrthDelta[:, None]
    return d * (vf[u] - vf[l]) + corr


def surface_integrate_full(l, u, ssf, patches, V):
    """fvc::surfaceIntegrate with its patch faces, fvcSurfaceIntegrate.C:43-76; patches: list of (faceCells, pssf)"""
    out = np.zeros((V.size,) + ssf.shape[1:])
    for f in range(l.size):
        out[l[f]] += ssf[f]
        out[u[f]] -= ssf[f]
    for fc, pssf in patches:
        for i, c in enumerate(fc):
            out[c] += pssf[i]
    return out / (V if ssf.ndim == 1 else V[:, None])


def source_minus_V_div(source, l, u, ffc, patches, V):
    """fvm.source() -= mesh.V()*fvc::div(faceFluxCorrection)().internalField(), gaussLaplacianSchemes.C:74-88 /
    gaussLaplacianScheme.C:187"""
    div = surface_integrate_full(l, u, ffc, patches, V)
    return source - (V if ffc.ndim == 1 else V[:, None]) * div


def face_scale(scale, field):
    """surfaceScalarField * surfaceField<Type>: s*v per face and component"""
    return scale * field if field.ndim == 1 else scale[:, None] * field


def tensor_gamma_factors(Sf, magSf, gamma):
    """gaussLaplacianScheme<Type, GType>::fvmLaplacian, gaussLaplacianScheme.C:165-173: Sn = Sf/magSf;
    SfGamma = Sf & gamma; SfGammaSn = SfGamma & Sn; SfGammaCorr = SfGamma - SfGammaSn*Sn.  gamma [n,6] (symmTensor,
    SymmTensorI.H operator&(Vector, SymmTensor)) or [n,9] (tensor)"""
    Sn = Sf / magSf[:, None]
    x, y, z = Sf[:, 0], Sf[:, 1], Sf[:, 2]
    if gamma.shape[1] == 6:
        xx, xy, xz, yy, yz, zz = (gamma[:, i] for i in range(6))
        SfGamma = np.stack([x * xx + y * xy + z * xz, x * xy + y * yy + z * yz, x * xz + y * yz + z * zz], axis=1)
    else:
        SfGamma = vec_dot_field(Sf, gamma)
    SfGammaSn = _dot(SfGamma, Sn)
    return SfGammaSn, SfGamma - SfGammaSn[:, None] * Sn


def gauss_div_faces(l, u, w, Sf, vf):
    """gaussDivScheme::fvcDiv, divSchemes/gaussDivScheme/gaussDivScheme.C:48-68: the face field
    Sf & linear.interpolate(vf) (vf vector -> scalar, tensor -> vector), internal faces"""
    return vec_dot_field(Sf, interpolate(l, u, w, vf))
