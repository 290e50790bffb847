"""GPU (-m gpu): finite-volume face stencils through the C ABI vs the numpy
restatement oracle/fv_oracle.py.  Bit-exact: the kernels gather per cell in the reference's face
order.  (fv oracle: parity unpinned, see its header.)"""
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
