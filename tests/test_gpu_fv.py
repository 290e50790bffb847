"""GPU (-m gpu): finite-volume face stencils through the C ABI vs the numpy
restatement oracle/fv_oracle.py AND directly vs the vectors of the reference's own libfiniteVolume
(tests/golden/fv_*.npz, produced by oracle/_ref/fv_driver).  Bit-exact: the kernels gather per cell
in the reference's face order."""
import os
import numpy as np
import pytest

from openfoam_amd import capi, cases

import fv_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("gen", [lambda: cases.box3d(9, 7, 5), lambda: cases.random_graph(400)],
                         ids=["box", "random"])
def test_fv_stencils_bitexact(ctx, gen):
    p = gen()
    nC = p["nCells"]
    l, u = p["lowerAddr"], p["upperAddr"]
    nF = l.size
    rng = np.random.RandomState(11)
    a = capi.Addressing(ctx, nC, l, u)
    lam, delta, gms, phi = rng.rand(nF), 0.5 + rng.rand(nF), 0.5 + rng.rand(nF), rng.randn(nF)
    V = 0.5 + rng.rand(nC)
    Sf = rng.randn(nF, 3)
    vf, vv = rng.randn(nC), rng.randn(nC, 3)
    ssf, ssv = rng.randn(nF), rng.randn(nF, 3)

    assert np.array_equal(a.interpolate(lam, vf), fo.interpolate(l, u, lam, vf))
    assert np.array_equal(a.interpolate(lam, vv), fo.interpolate(l, u, lam, vv))
    assert np.array_equal(a.surfaceIntegrate(ssf, V), fo.surface_integrate(l, u, ssf, V))
    assert np.array_equal(a.surfaceIntegrate(ssv, V), fo.surface_integrate(l, u, ssv, V))
    assert np.array_equal(a.gaussGrad(Sf, ssf, V), fo.gauss_grad(l, u, Sf, ssf, V))
    assert np.array_equal(a.snGrad(delta, vf), fo.sn_grad(l, u, delta, vf))
    d, up = a.fvmLaplacian(delta, gms)
    d0, up0 = fo.fvm_laplacian(nC, l, u, delta, gms)
    assert np.array_equal(up, up0) and np.array_equal(d, d0)
    d, up, lo = a.fvmDiv(lam, phi)
    d0, up0, lo0 = fo.fvm_div(nC, l, u, lam, phi)
    assert np.array_equal(up, up0) and np.array_equal(lo, lo0) and np.array_equal(d, d0)
    a.close()


@pytest.mark.parametrize("name", ["fv_box_7x6x5", "fv_box_12x3x9"])
def test_fv_kernels_against_reference_vectors(ctx, name):
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")))
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    w, d, V, Sf = g["ref_weights"], g["ref_deltaCoeffs"], g["ref_V"], g["ref_Sf"]
    a = capi.Addressing(ctx, nC, l, u)
    eq = np.array_equal
    assert eq(a.interpolate(w, g["vf"]), g["ref_interpolate_s"])
    assert eq(a.interpolate(w, g["U"]), g["ref_interpolate_v"])
    assert eq(a.interpolate(g["ref_upwindWeights"], g["vf"]), g["ref_interpolate_upwind"])
    assert eq(a.surfaceIntegrate(g["phi"], V), g["ref_surfaceIntegrate_s"])
    assert eq(a.surfaceIntegrate(g["ref_phiU"], V), g["ref_surfaceIntegrate_v"])
    assert eq(a.gaussGrad(Sf, g["ref_interpolate_s"], V), g["ref_gaussGrad"])
    assert eq(a.snGrad(d, g["vf"]), g["ref_snGrad"])
    diag, upper = a.fvmLaplacian(d, g["ref_gammaMagSf"])
    assert eq(upper, g["ref_laplacian_upper"]) and eq(diag, g["ref_laplacian_diag"])
    for kind, wk in (("linear", w), ("upwind", g["ref_upwindWeights"])):
        diag, upper, lower = a.fvmDiv(wk, g["phi"])
        assert eq(lower, g["ref_div_%s_lower" % kind]) and eq(upper, g["ref_div_%s_upper" % kind])
        assert eq(diag, g["ref_div_%s_diag" % kind])
    # a30: the geometric agglomeration weights from Sf alone, on the device, vs the reference's
    # mag(cmptMultiply(mesh.Sf()/sqrt(mesh.magSf()), (1, 1.01, 1.02))) (faceAreaPairGAMGAgglomeration.C:48-73)
    assert eq(a.set_face_areas(Sf), g["ref_faceAreaPairWeights"])
    a.close()


def test_assembled_laplacian_solves(ctx, oracle):
    """stencil -> matrix -> solve: fvm::laplacian coefficients assembled on the device feed PCG."""
    p = cases.box3d(10)
    nC, l, u = p["nCells"], p["lowerAddr"], p["upperAddr"]
    a = capi.Addressing(ctx, nC, l, u, p["faceWeights"])
    delta = np.ones(l.size)
    gms = -p["upper"]                      # gamma*magSf so that upper = -(...) as in cases.box3d
    d, up = a.fvmLaplacian(delta, -gms)
    assert np.array_equal(up, p["upper"])
    d[0] *= 2.0
    np.testing.assert_allclose(d, p["diag"], rtol=1e-14)
    m = capi.Matrix(a)
    m.set_coeffs(d, up)
    x, perf = m.solve(p["psi"], p["source"], solver="PCG", preconditioner="DIC", tolerance=1e-9, relTol=0)
    p2 = dict(p, diag=d, upper=up)
    xo, po = oracle.System(p2).solve(p["psi"], p["source"], solver="PCG", precond="DIC", tolerance=1e-9, relTol=0)
    assert perf["nIterations"] == po["nIterations"]
    m.close(); a.close()


@pytest.mark.parametrize("name", ["fvglue_box_6x5x4_cyclic", "fvglue_box_3x9x2"])
def test_fvmatrix_glue_against_reference_vectors(ctx, name):
    """ldu_fvm_addBoundaryDiag/addBoundarySource/relax/setReference/A/H/flux vs what the reference's own
    fvMatrix computed (fixedValue, zeroGradient and cyclic patches): bit-exact."""
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")))
    nP = int(g["nPatches"][0])
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    a = capi.Addressing(ctx, nC, l, u)
    fcs = [g["p%d_faceCells" % p] for p in range(nP)]
    cp = [int(g["p%d_coupled" % p][0]) for p in range(nP)]
    cat = lambda key: np.concatenate([g["p%d_%s" % (p, key)] for p in range(nP)])
    iC, bC, pnf = cat("internalCoeffs"), cat("boundaryCoeffs"), cat("pnf")
    B = capi.FvBoundary(a, fcs, cp)
    eq = np.array_equal
    assert eq(B.addBoundaryDiag(iC, g["diag"]), g["ref_addBoundaryDiag"])
    assert eq(B.addBoundarySource(bC, pnf, g["source"]), g["ref_addBoundarySource"])
    assert eq(B.addBoundarySource(bC, pnf, g["source"], couples=False), g["ref_addBoundarySource_nocouples"])
    assert eq(B.A(iC, g["diag"], g["V"]), g["ref_A"])
    assert eq(B.H(iC, bC, pnf, g["upper"], g["lower"], g["psi"], g["source"], g["V"]), g["ref_H"])
    fi, fb = B.flux(iC, bC, pnf, g["upper"], g["lower"], g["psi"])
    assert eq(fi, g["ref_flux_internal"]) and eq(fb, cat("ref_flux"))
    d, s = B.relax(0.7, iC, bC, g["upper"], g["lower"], g["psi"], g["diag"], g["source"])
    assert eq(d, g["ref_relax_diag"]) and eq(s, g["ref_relax_source"])
    d, s = B.setReference(5, 1.3, g["diag"], g["source"])
    assert eq(d, g["ref_setReference_diag"]) and eq(s, g["ref_setReference_source"])
    d, s = B.setReference(-1, 1.3, g["diag"], g["source"])     # needReference false / no cell: no-op
    assert eq(d, g["diag"]) and eq(s, g["source"])
    B.close(); a.close()


def test_fvmatrix_glue_device_resident(ctx):
    """relax with every array resident in HBM (raw hipMalloc pointers handed to the C ABI: no staging)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")   # the runtime libldugpu.so already loaded
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fvglue_box_3x9x2.npz")))
    nP = int(g["nPatches"][0])
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    a = capi.Addressing(ctx, nC, l, u)
    B = capi.FvBoundary(a, [g["p%d_faceCells" % p] for p in range(nP)], [0] * nP)
    cat = lambda key: np.concatenate([g["p%d_%s" % (p, key)] for p in range(nP)])
    bufs = []

    def dev(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(x.nbytes)) == 0
        assert hip.hipMemcpy(p, x.ctypes.data_as(C.c_void_p), C.c_size_t(x.nbytes), 1) == 0   # host -> device
        bufs.append(p)
        return p

    def host(p, n):
        out = np.zeros(n)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes), 2) == 0
        return out

    iC, bC = dev(cat("internalCoeffs")), dev(cat("boundaryCoeffs"))
    diag, source, up, lo, psi = dev(g["diag"]), dev(g["source"]), dev(g["upper"]), dev(g["lower"]), dev(g["psi"])
    capi._chk(capi.lib().ldu_fvm_relax(B.h, C.c_double(0.7), iC, bC, up, lo, psi, diag, source))
    assert np.array_equal(host(diag, nC), g["ref_relax_diag"])
    assert np.array_equal(host(source, nC), g["ref_relax_source"])
    for p in bufs:
        hip.hipFree(p)
    B.close(); a.close()


def test_end_to_end_assembly_to_solve_against_reference(ctx):
    """Device path: ldu_fvm_addBoundaryDiag + ldu_fvm_addBoundarySource(couples=false) + ldu_solve, i.e.
    fvScalarMatrix::solveSegregated (fvScalarMatrix.C:136-183), against the reference's own
    fvScalarMatrix::solve with GAMG/faceAreaPair (real libfiniteVolume agglomerator) and PCG/DIC."""
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fvsolve_box_14x12x10.npz")))
    nP = int(g["nPatches"][0])
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    a = capi.Addressing(ctx, nC, l, u, g["faceAreaPairWeights"])
    B = capi.FvBoundary(a, [g["p%d_faceCells" % p] for p in range(nP)], [0] * nP)
    cat = lambda key: np.concatenate([g["p%d_%s" % (p, key)] for p in range(nP)])
    diag = B.addBoundaryDiag(cat("internalCoeffs"), g["diag"])
    source = B.addBoundarySource(cat("boundaryCoeffs"), None, g["source"], couples=False)
    m = capi.Matrix(a)
    m.set_coeffs(diag, g["upper"])
    x, perf = m.solve(np.zeros(nC), source, solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                      nCellsInCoarsestLevel=10, mergeLevels=1, tolerance=1e-10, relTol=0)
    r = g["ref_gamg_perf"]
    assert perf["nIterations"] == int(r[2]) and perf["converged"]
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-9 * np.max(np.abs(g["ref_gamg_psi"]))
    x, perf = m.solve(np.zeros(nC), source, solver="PCG", preconditioner="DIC", tolerance=1e-10, relTol=0)
    r = g["ref_pcg_perf"]
    assert perf["nIterations"] == int(r[2])
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_pcg_psi"])) <= 1e-9 * np.max(np.abs(g["ref_pcg_psi"]))
    m.close(); B.close(); a.close()


def test_vector_fvmatrix_glue_against_reference_vectors(ctx):
    """ldu_fvm_*V (fvMatrix<vector>) vs the reference's own fvVectorMatrix: bit-exact."""
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fvglueV_box_5x6x4_cyclic.npz")))
    nP = int(g["nPatches"][0])
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    a = capi.Addressing(ctx, nC, l, u)
    cat = lambda key: np.concatenate([g["p%d_%s" % (p, key)] for p in range(nP)])
    iC, bC, pnf = cat("internalCoeffs"), cat("boundaryCoeffs"), cat("pnf")
    B = capi.FvBoundary(a, [g["p%d_faceCells" % p] for p in range(nP)], [int(g["p%d_coupled" % p][0]) for p in range(nP)])
    eq = np.array_equal
    for k in range(3):
        assert eq(B.addBoundaryDiagCmpt(iC, k, g["diag"]), g["ref_addBoundaryDiag%d" % k])
    assert eq(B.addBoundarySourceV(bC, pnf, g["source"]), g["ref_addBoundarySource"])
    assert eq(B.addBoundarySourceV(bC, pnf, g["source"], couples=False), g["ref_addBoundarySource_nocouples"])
    assert eq(B.AV(iC, g["diag"], g["V"]), g["ref_A"])
    assert eq(B.HV(iC, bC, pnf, g["upper"], g["lower"], g["psi"], g["source"], g["V"]), g["ref_H"])
    d, s = B.relaxV(0.7, iC, bC, g["upper"], g["lower"], g["psi"], g["diag"], g["source"])
    assert eq(d, g["ref_relax_diag"]) and eq(s, g["ref_relax_source"])
    B.close(); a.close()


@pytest.mark.parametrize("name", ["fv_box_7x6x5", "fv_box_12x3x9"])
def test_higher_order_schemes_against_reference_vectors(ctx, name):
    """ldu_fv_linearUpwindCorrection / ldu_fvc_cellLimitedGrad vs the reference's own linearUpwind<scalar>
    and cellLimitedGrad<scalar> (k = 1 and 0.5): bit-exact."""
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")))
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    nP = int(g["ref_nPatches"][0])
    a = capi.Addressing(ctx, nC, l, u)
    C, Cf, g0 = g["ref_C"], g["ref_Cf"], g["ref_gaussLinearGrad"]
    assert np.array_equal(a.linearUpwindCorrection(g["phi"], C, Cf, g0), g["ref_linearUpwind_correction"])
    B = capi.FvBoundary(a, [g["ref_p%d_faceCells" % p] for p in range(nP)], [0] * nP)
    bVal = np.concatenate([g["ref_p%d_value" % p] for p in range(nP)])
    bCf = np.concatenate([g["ref_p%d_Cf" % p] for p in range(nP)])
    assert np.array_equal(B.cellLimitedGrad(1.0, g["vf"], bVal, C, Cf, bCf, g0), g["ref_cellLimitedGrad_k1"])
    assert np.array_equal(B.cellLimitedGrad(0.5, g["vf"], bVal, C, Cf, bCf, g0), g["ref_cellLimitedGrad_k05"])
    # vector forms (linearUpwindV, cellLimitedGrad<vector>): what motorBike's fvSchemes selects for U
    gU = g["ref_gaussLinearGradU"]
    corr = a.linearUpwindVCorrection(g["phi"], g["ref_weights"], g["U"], C, Cf, gU)
    assert np.array_equal(corr, g["ref_linearUpwindV_correction"]) and np.count_nonzero(corr)
    bValU = np.concatenate([g["ref_p%d_valueU" % p] for p in range(nP)])
    lim1 = B.cellLimitedGradV(1.0, g["U"], bValU, C, Cf, bCf, gU)
    assert np.array_equal(lim1, g["ref_cellLimitedGradV_k1"]) and not np.array_equal(lim1, gU)
    assert np.array_equal(B.cellLimitedGradV(0.5, g["U"], bValU, C, Cf, bCf, gU), g["ref_cellLimitedGradV_k05"])
    # the basic gradient of U that feeds those schemes: linear interpolation to the faces, then Gauss with the
    # patch faces (their area vectors come from the device geometry of the generating mesh)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import fv_case
    import make_fv_golden
    nx, ny, nz, seed = make_fv_golden.CASES[name]
    mesh = fv_case.box_mesh(nx, ny, nz, seed=seed)
    start, pts = capi.faces_csr(mesh["faces"])
    _, SfAll, _, _ = capi.mesh_geometry(ctx, mesh["points"], start, pts, mesh["owner"], mesh["neighbour"], nC)
    nI = l.size
    assert np.array_equal(SfAll[:nI], g["ref_Sf"])
    Uf = a.interpolate(g["ref_weights"], g["U"])
    gradU = B.gaussGradFull(g["ref_Sf"], Uf, SfAll[nI:], bValU, g["ref_V"])
    assert np.array_equal(gradU, gU)
    # the `bounded` wrapper: fvmDiv - fvm::Sp(fvc::surfaceIntegrate(phi)) with non-zero boundary fluxes
    bPhi = np.concatenate([g["ref_p%d_phi" % p] for p in range(nP)])
    bd = B.boundedSp(g["phi"], bPhi, g["ref_V"], g["ref_div_upwind_diag_bphi"])
    assert np.array_equal(bd, g["ref_div_bounded_upwind_diag"]) and not np.array_equal(bd, g["ref_div_upwind_diag_bphi"])
    B.close(); a.close()
