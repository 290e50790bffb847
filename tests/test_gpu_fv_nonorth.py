"""GPU (-m gpu): the HIP kernels of ldu_fvschemes.hip through the C ABI - nonOrthDeltaCoeffs / nonOrthCorrectionVectors,
correctedSnGrad (correction, snGrad), the corrected gaussLaplacianScheme (scalar, symmTensor, tensor gamma; fvm source
and fvc), gaussDivScheme::fvcDiv of vector and tensor fields, interpolation on patch faces (coupled and not) and
gaussGrad's boundary correction - bit-exact against the reference's own classes (tests/golden/fvnonorth_*.npz)."""
import numpy as np
import pytest

from openfoam_amd import capi

import nonorth_common as nc

pytestmark = pytest.mark.gpu


class GpuBackend:
    def __init__(self, ctx):
        self.ctx = ctx

    def setup(self, nC, l, u, faceCells, coupled):
        self.nC, self.l, self.u = nC, l, u
        self.a = capi.Addressing(self.ctx, nC, l, u)
        self.b = capi.FvBoundary(self.a, faceCells, [int(c) for c in coupled])

    def teardown(self):
        self.b.close(); self.a.close()

    def nonorth_factors(self, C, Sf, magSf):
        return capi.mesh_nonorth_factors(self.ctx, self.nC, self.l, self.u, Sf, magSf, C)

    def nonorth_factors_patch(self, Sf, magSf, delta, coupled):
        return capi.mesh_patch_nonorth_factors(self.ctx, Sf, magSf, delta, coupled)

    def interpolate(self, w, vf):
        return self.a.interpolate(w, vf)

    def interpolate_boundary(self, wB, vf, pnfB, valuesB):
        nc_ = 1 if vf.ndim == 1 else vf.shape[1]
        if valuesB is None:
            valuesB = np.zeros((self.b.n, nc_))
        return self.b.interpolateBoundary(wB, vf, pnfB, valuesB)

    def gauss_grad_full(self, Sf, sf, SfB, sfB, V):
        return self.b.gaussGradFull(Sf, sf, SfB, sfB, V)

    def gauss_grad_boundary(self, nfB, grad, snB, gb):
        return self.b.gaussGradBoundary(nfB, grad, snB, gb)

    def interpolate_dot(self, vec, w, field):
        return self.a.interpolateDot(vec, w, field)

    def face_dot(self, vec, field):
        return capi.fv_face_dot(self.ctx, vec, field)

    def face_scale(self, scale, field, accumulate_into=None):
        return capi.fv_face_scale(self.ctx, scale, field, accumulate_into)

    def corrected_sn_grad(self, nod, vf, corr):
        return self.a.correctedSnGrad(nod, vf, corr)

    def fvm_laplacian(self, delta, gms):
        return self.a.fvmLaplacian(delta, gms)

    def source_minus_V_div(self, source, ffc, ffcB, V):
        return self.a.sourceMinusVDiv(source, ffc, V, self.b, ffcB)

    def surface_integrate_full(self, ssf, ssfB, V):
        return self.a.surfaceIntegrateFull(ssf, V, self.b, ssfB)

    def tensor_gamma_factors(self, Sf, magSf, gamma):
        return capi.fv_tensor_gamma_factors(self.ctx, Sf, magSf, gamma)


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", nc.CASES)
def test_nonorth_chain_hip_matches_reference(ctx, name):
    assert nc.run_chain(name, GpuBackend(ctx)) == []


def test_device_resident_fields(ctx):
    """the same kernels on device pointers (torch tensors): nothing is staged through the host"""
    torch = pytest.importorskip("torch")
    g, P = nc.load("fvnonorth_prism_5x4x3")
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    a = capi.Addressing(ctx, nC, l, u)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    cv, w, gT = dev(nc.rs(g["nonOrthCorrectionVectors"], 3)), dev(g["weights"]), dev(nc.rs(g["ref_gradT"], 3))
    out = torch.zeros(l.size, dtype=torch.float64, device="cuda")
    capi._chk(capi.lib().ldu_fv_interpolateDot(a.h, 3, capi._ptr(cv), capi._ptr(w), capi._ptr(gT), capi._ptr(out)))
    assert np.array_equal(out.cpu().numpy(), g["ref_snGradCorrection_T"])
    a.close()
