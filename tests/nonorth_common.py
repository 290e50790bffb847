"""The non-orthogonal / gaussDiv chain composed the way a caller composes it, once
for any backend: the numpy restatement (oracle/fv_oracle.py, pinned here against the reference's vectors) and the
HIP kernels behind the C ABI (tests/test_gpu_fv_nonorth.py).  Every comparison is bit-exact against what the
reference's own classes produced (tests/golden/fvnonorth_*.npz, oracle/fv_driver.C mode nonorth)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["fvnonorth_box_6x5x4", "fvnonorth_box_5x4x6_cyclic", "fvnonorth_prism_5x4x3"]


def rs(a, k):
    return a.reshape(-1, k)


def load(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    nP = int(g["nPatches"][0])
    P = []
    for p in range(nP):
        q = lambda k, p=p: g["p%d_%s" % (p, k)]
        P.append(dict(fc=q("faceCells").astype(np.int32), coupled=bool(q("coupled")[0]), Sf=rs(q("Sf"), 3),
                      magSf=q("magSf"), delta=rs(q("delta"), 3), nf=rs(q("nf"), 3), w=q("weights"), T=q("T"),
                      U=rs(q("U"), 3), Tpnf=q("T_pnf"), Upnf=rs(q("U_pnf"), 3), Tsn=q("T_snGrad"),
                      Usn=rs(q("U_snGrad"), 3), gamma=q("gamma"), q=q))
    return g, P


def cat(P, f):
    xs = [f(pp) for pp in P]
    return np.concatenate(xs) if xs else np.zeros(0)


def outer(a, b):
    return np.stack([a[:, i] * b[:, j] for i in range(3) for j in range(3)], axis=1)


def run_chain(name, be):
    """be: backend with the operations of include/ldugpu.h (see OracleBackend below).  Returns the list of failed
    comparisons (empty = every reference vector reproduced bit for bit)."""
    g, P = load(name)
    eq = np.array_equal
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    C, Sf, magSf, w, V = rs(g["C"], 3), rs(g["Sf"], 3), g["magSf"], g["weights"], g["V"]
    T, U = g["T"], rs(g["U"], 3)
    bad = []

    def chk(what, got, ref):
        if not eq(np.asarray(got).reshape(-1), np.asarray(ref).reshape(-1)):
            bad.append(what)

    be.setup(nC, l, u, [pp["fc"] for pp in P], [pp["coupled"] for pp in P])
    coupledB = cat(P, lambda pp: np.full(pp["fc"].size, pp["coupled"]))
    # geometry
    nod, cv = be.nonorth_factors(C, Sf, magSf)
    chk("nonOrthDeltaCoeffs", nod, g["nonOrthDeltaCoeffs"])
    chk("nonOrthCorrectionVectors", cv, g["nonOrthCorrectionVectors"])
    nod2, _ = be.nonorth_factors(C, Sf, None)                   # magSf computed from Sf
    chk("nonOrthDeltaCoeffs(magSf=None)", nod2, g["nonOrthDeltaCoeffs"])
    for p, pp in enumerate(P):
        a, b = be.nonorth_factors_patch(pp["Sf"], pp["magSf"], pp["delta"], pp["coupled"])
        chk("p%d nonOrthDeltaCoeffs" % p, a, pp["q"]("nonOrthDeltaCoeffs"))
        chk("p%d nonOrthCorrectionVectors" % p, b, pp["q"]("nonOrthCorrectionVectors"))
        pp["cv"], pp["nod"] = b, a
    # a35 patch half
    wB = cat(P, lambda pp: pp["w"])
    iT = be.interpolate_boundary(wB, T, cat(P, lambda pp: pp["Tpnf"]), cat(P, lambda pp: pp["T"]))
    iU = be.interpolate_boundary(wB, U, cat(P, lambda pp: pp["Upnf"]), cat(P, lambda pp: pp["U"]))
    chk("interpolate_T patches", iT, cat(P, lambda pp: pp["q"]("ref_interpolate_T")))
    chk("interpolate_U patches", iU, cat(P, lambda pp: rs(pp["q"]("ref_interpolate_U"), 3)))
    # a34: Gauss linear gradients with their patch faces, then the boundary correction
    SfB = cat(P, lambda pp: pp["Sf"])
    gT = be.gauss_grad_full(Sf, be.interpolate(w, T), SfB, iT, V)
    gU = be.gauss_grad_full(Sf, be.interpolate(w, U), SfB, iU, V)
    chk("grad(T)", gT, g["ref_gradT"])
    chk("grad(U)", gU, g["ref_gradU"])
    nfB = cat(P, lambda pp: pp["nf"])
    gTb = be.interpolate_boundary(wB, gT, cat(P, lambda pp: rs(pp["q"]("gradT_pnf"), 3)), None)
    gUb = be.interpolate_boundary(wB, gU, cat(P, lambda pp: rs(pp["q"]("gradU_pnf"), 9)), None)
    gTb = be.gauss_grad_boundary(nfB, gT, cat(P, lambda pp: pp["Tsn"]), gTb)
    gUb = be.gauss_grad_boundary(nfB, gU, cat(P, lambda pp: pp["Usn"]), gUb)
    chk("grad(T) patches", gTb, cat(P, lambda pp: rs(pp["q"]("ref_gradT"), 3)))
    chk("grad(U) patches", gUb, cat(P, lambda pp: rs(pp["q"]("ref_gradU"), 9)))
    # a36
    cT = be.interpolate_dot(cv, w, gT)
    cU = be.interpolate_dot(cv, w, gU)
    chk("correctedSnGrad::correction(T)", cT, g["ref_snGradCorrection_T"])
    chk("correctedSnGrad::correction(U)", cU, g["ref_snGradCorrection_U"])
    chk("snGrad(T) corrected", be.corrected_sn_grad(nod, T, cT), g["ref_correctedSnGrad_T"])
    chk("snGrad(U) corrected", be.corrected_sn_grad(nod, U, cU), g["ref_correctedSnGrad_U"])
    cvB = cat(P, lambda pp: pp["cv"])
    cTb = be.face_dot(cvB, gTb)
    cUb = be.face_dot(cvB, gUb)
    chk("correction(T) patches", cTb, cat(P, lambda pp: pp["q"]("ref_snGradCorrection_T")))
    chk("correction(U) patches", cUb, cat(P, lambda pp: rs(pp["q"]("ref_snGradCorrection_U"), 3)))
    # a37 scalar gamma: fvm
    gam = g["gamma"]
    gms = gam * magSf                                  # surfaceScalarField product of the caller (gamma*magSf)
    gmsB = cat(P, lambda pp: pp["gamma"] * pp["magSf"])
    diag, upper = be.fvm_laplacian(nod, gms)
    chk("laplacian upper", upper, g["ref_lap_upper"])
    chk("laplacian diag", diag, g["ref_lap_diag"])
    ffc = be.face_scale(gms, cT)
    chk("faceFluxCorrection", ffc, g["ref_lap_faceFluxCorrection"])
    ffcB = be.face_scale(gmsB, cTb)
    chk("faceFluxCorrection patches", ffcB, cat(P, lambda pp: pp["q"]("ref_lap_faceFluxCorrection")))
    chk("laplacian source", be.source_minus_V_div(np.zeros(nC), ffc, ffcB, V), g["ref_lap_source"])
    chk("laplacian(U) source", be.source_minus_V_div(np.zeros((nC, 3)), be.face_scale(gms, cU), be.face_scale(gmsB, cUb), V),
        g["ref_lapU_source"])
    # fvc::laplacian(gamma, T) = div((gamma*snGrad)*magSf)
    sg = be.corrected_sn_grad(nod, T, cT)
    sgB = cat(P, lambda pp: (pp["nod"] * (pp["Tpnf"] - T[pp["fc"]]) if pp["coupled"] else pp["Tsn"])) + cTb
    chk("fvc::laplacian", be.surface_integrate_full((gam * sg) * magSf, (cat(P, lambda pp: pp["gamma"]) * sgB) * cat(P, lambda pp: pp["magSf"]), V),
        g["ref_fvcLaplacian"])
    # a37 tensor gamma
    for key, k in (("S", 6), ("T", 9)):
        sn, sc = be.tensor_gamma_factors(Sf, magSf, rs(g["gamma" + key], k))
        diag, upper = be.fvm_laplacian(nod, sn)
        chk("laplacian%s upper" % key, upper, g["ref_lap%s_upper" % key])
        chk("laplacian%s diag" % key, diag, g["ref_lap%s_diag" % key])
        snB, scB = be.tensor_gamma_factors(SfB, cat(P, lambda pp: pp["magSf"]), cat(P, lambda pp: rs(pp["q"]("gamma" + key), k)))
        tffc = be.face_scale(sn, cT, accumulate_into=be.interpolate_dot(sc, w, gT))
        tffcB = be.face_scale(snB, cTb, accumulate_into=be.face_dot(scB, gTb))
        chk("laplacian%s source" % key, be.source_minus_V_div(np.zeros(nC), tffc, tffcB, V), g["ref_lap%s_source" % key])
        if key == "S":
            chk("fvc::laplacianS", be.surface_integrate_full(sn * sg + be.interpolate_dot(sc, w, gT), snB * sgB + be.face_dot(scB, gTb), V),
                g["ref_fvcLaplacianS"])
    # a39
    chk("gaussDiv(U)", be.surface_integrate_full(be.interpolate_dot(Sf, w, U), be.face_dot(SfB, iU), V), g["ref_divU"])
    chk("gaussDiv(grad(U))", be.surface_integrate_full(be.interpolate_dot(Sf, w, gU), be.face_dot(SfB, gUb), V), g["ref_divGradU"])
    be.teardown()
    return bad


class OracleBackend:
    """oracle/fv_oracle.py behind the backend interface"""

    def __init__(self, fo):
        self.fo = fo

    def setup(self, nC, l, u, faceCells, coupled):
        self.nC, self.l, self.u = nC, l, u
        self.fc, self.coupled = faceCells, coupled
        self.off = np.concatenate([[0], np.cumsum([len(x) for x in faceCells])]).astype(int)

    def teardown(self):
        pass

    def _split(self, a):
        return [a[self.off[i]:self.off[i + 1]] for i in range(len(self.fc))]

    def nonorth_factors(self, C, Sf, magSf):
        if magSf is None:
            magSf = np.sqrt(Sf[:, 0] * Sf[:, 0] + Sf[:, 1] * Sf[:, 1] + Sf[:, 2] * Sf[:, 2]) + 1e-300
        return self.fo.nonorth_factors(self.l, self.u, C, Sf, magSf)

    def nonorth_factors_patch(self, Sf, magSf, delta, coupled):
        return self.fo.nonorth_factors_patch(Sf, magSf, delta, coupled)

    def interpolate(self, w, vf):
        return self.fo.interpolate(self.l, self.u, w, vf)

    def interpolate_boundary(self, wB, vf, pnfB, valuesB):
        out = []
        for i, fc in enumerate(self.fc):
            s = slice(self.off[i], self.off[i + 1])
            if self.coupled[i]:
                out.append(self.fo.interpolate_patch(wB[s], vf, fc, pnfB[s], None, True))
            else:
                out.append(valuesB[s].copy() if valuesB is not None else np.zeros((len(fc),) + vf.shape[1:]))
        return np.concatenate(out) if out else np.zeros(0)

    def gauss_grad_full(self, Sf, sf, SfB, sfB, V):
        if sf.ndim == 1:
            ssf, ssfB = Sf * sf[:, None], SfB * sfB[:, None]
        else:
            ssf, ssfB = outer(Sf, sf), outer(SfB, sfB)
        return self.fo.surface_integrate_full(self.l, self.u, ssf, list(zip(self.fc, self._split(ssfB))), V)

    def gauss_grad_boundary(self, nfB, grad, snB, gb):
        out = np.array(gb, copy=True)
        for i, fc in enumerate(self.fc):
            if self.coupled[i]:
                continue
            s = slice(self.off[i], self.off[i + 1])
            out[s] = self.fo.gauss_grad_boundary(nfB[s], grad[fc], snB[s])
        return out

    def interpolate_dot(self, vec, w, field):
        return self.fo.vec_dot_field(vec, self.fo.interpolate(self.l, self.u, w, field))

    def face_dot(self, vec, field):
        return self.fo.vec_dot_field(vec, field)

    def face_scale(self, scale, field, accumulate_into=None):
        v = self.fo.face_scale(scale, field)
        return v if accumulate_into is None else accumulate_into + v

    def corrected_sn_grad(self, nod, vf, corr):
        return self.fo.corrected_sn_grad(self.l, self.u, nod, vf, corr)

    def fvm_laplacian(self, delta, gms):
        return self.fo.fvm_laplacian(self.nC, self.l, self.u, delta, gms)

    def source_minus_V_div(self, source, ffc, ffcB, V):
        return self.fo.source_minus_V_div(source, self.l, self.u, ffc, list(zip(self.fc, self._split(ffcB))), V)

    def surface_integrate_full(self, ssf, ssfB, V):
        return self.fo.surface_integrate_full(self.l, self.u, ssf, list(zip(self.fc, self._split(ssfB))), V)

    def tensor_gamma_factors(self, Sf, magSf, gamma):
        return self.fo.tensor_gamma_factors(Sf, magSf, gamma)
