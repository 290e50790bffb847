"""CPU: oracle/fv_oracle.py (numpy restatement of the fv face stencils)
against vectors produced by the REFERENCE's own libfiniteVolume (tests/golden/make_fv_golden.py ->
oracle/_ref/fv_driver).  Bit-exact: same operations in the same order."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fv_oracle  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
NAMES = ["fv_box_7x6x5", "fv_box_12x3x9"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.mark.parametrize("name", NAMES)
def test_fv_oracle_matches_reference(name):
    g = load(name)
    l, u, nC = g["lowerAddr"], g["upperAddr"], int(g["nCells"])
    w, d, V, Sf = g["ref_weights"], g["ref_deltaCoeffs"], g["ref_V"], g["ref_Sf"]
    eq = np.array_equal
    assert eq(fv_oracle.interpolate(l, u, w, g["vf"]), g["ref_interpolate_s"])
    assert eq(fv_oracle.interpolate(l, u, w, g["U"]), g["ref_interpolate_v"])
    assert eq(fv_oracle.interpolate(l, u, g["ref_upwindWeights"], g["vf"]), g["ref_interpolate_upwind"])
    assert eq(fv_oracle.upwind_weights(g["phi"]), g["ref_upwindWeights"])
    assert eq(fv_oracle.surface_integrate(l, u, g["phi"], V), g["ref_surfaceIntegrate_s"])
    assert eq(fv_oracle.surface_integrate(l, u, g["ref_phiU"], V), g["ref_surfaceIntegrate_v"])
    assert eq(fv_oracle.gauss_grad(l, u, Sf, g["ref_interpolate_s"], V), g["ref_gaussGrad"])
    assert eq(fv_oracle.sn_grad(l, u, d, g["vf"]), g["ref_snGrad"])
    diag, upper = fv_oracle.fvm_laplacian(nC, l, u, d, g["ref_gammaMagSf"])
    assert eq(upper, g["ref_laplacian_upper"]) and eq(diag, g["ref_laplacian_diag"])
    for kind, wk in (("linear", w), ("upwind", g["ref_upwindWeights"])):
        diag, upper, lower = fv_oracle.fvm_div(nC, l, u, wk, g["phi"])
        assert eq(lower, g["ref_div_%s_lower" % kind]), kind
        assert eq(upper, g["ref_div_%s_upper" % kind]), kind
        assert eq(diag, g["ref_div_%s_diag" % kind]), kind
    assert eq(fv_oracle.face_area_pair_weights(Sf, g["ref_magSf"]), g["ref_faceAreaPairWeights"])


def test_fixture_is_live_when_the_reference_build_exists(tmp_path):
    """With oracle/_ref/fv_driver present (this container), regenerate one fixture and require identity."""
    import fv_case
    if not fv_case.driver_available():
        pytest.skip("oracle/_ref/fv_driver not built (oracle/build_ref_fv.sh)")
    sys.path.insert(0, GOLDEN)
    import make_fv_golden
    fresh = make_fv_golden.generate("fv_box_7x6x5")
    old = load("fv_box_7x6x5")
    for k in old:
        assert np.array_equal(np.asarray(fresh[k]), old[k]), k


GLUE = ["fvglue_box_6x5x4_cyclic", "fvglue_box_3x9x2"]


def glue_patches(g):
    return [dict(faceCells=g["p%d_faceCells" % p], internalCoeffs=g["p%d_internalCoeffs" % p],
                 boundaryCoeffs=g["p%d_boundaryCoeffs" % p], coupled=bool(g["p%d_coupled" % p][0]),
                 pnf=g["p%d_pnf" % p]) for p in range(int(g["nPatches"][0]))]


@pytest.mark.parametrize("name", GLUE)
def test_fvmatrix_glue_oracle_matches_reference(name):
    """fvMatrix::addBoundaryDiag/addBoundarySource/A/H/flux/relax/setReference as executed by the
    reference's own fvMatrix (fixedValue, zeroGradient and cyclic patches) - bit-exact."""
    g = load(name)
    P = glue_patches(g)
    l, u = g["lowerAddr"], g["upperAddr"]
    eq = np.array_equal
    assert any(p["coupled"] for p in P) == ("cyclic" in name)
    assert eq(fv_oracle.add_boundary_diag(g["diag"], P), g["ref_addBoundaryDiag"])
    assert eq(fv_oracle.add_boundary_source(g["source"], P), g["ref_addBoundarySource"])
    assert eq(fv_oracle.add_boundary_source(g["source"], P, couples=False), g["ref_addBoundarySource_nocouples"])
    assert eq(fv_oracle.fvm_A(g["diag"], P, g["V"]), g["ref_A"])
    assert eq(fv_oracle.fvm_H(g["diag"], g["source"], l, u, g["upper"], g["lower"], g["psi"], P, g["V"]), g["ref_H"])
    fi, fb = fv_oracle.fvm_flux(l, u, g["upper"], g["lower"], g["psi"], P)
    assert eq(fi, g["ref_flux_internal"])
    for p in range(len(P)):
        assert eq(fb[p], g["p%d_ref_flux" % p]), p
    d, s = fv_oracle.relax(0.7, g["diag"], g["source"], l, u, g["upper"], g["lower"], g["psi"], P)
    assert eq(d, g["ref_relax_diag"]) and eq(s, g["ref_relax_source"])
    d, s = fv_oracle.set_reference(5, 1.3, g["diag"], g["source"])
    assert eq(d, g["ref_setReference_diag"]) and eq(s, g["ref_setReference_source"])


def solve_problem(g):
    """what fvScalarMatrix::solveSegregated hands to lduMatrix::solver (fvScalarMatrix.C:152-167)"""
    nP = int(g["nPatches"][0])
    P = [dict(faceCells=g["p%d_faceCells" % p], internalCoeffs=g["p%d_internalCoeffs" % p],
              boundaryCoeffs=g["p%d_boundaryCoeffs" % p], coupled=False, pnf=None) for p in range(nP)]
    diag = fv_oracle.add_boundary_diag(g["diag"], P)
    source = fv_oracle.add_boundary_source(g["source"], P, couples=False)
    return dict(nCells=int(g["nCells"]), lowerAddr=g["lowerAddr"], upperAddr=g["upperAddr"], diag=diag,
                upper=g["upper"], source=source, psi=np.zeros(int(g["nCells"])),
                faceWeights=g["ref_faceAreaPairWeights"] if "ref_faceAreaPairWeights" in g else g["faceAreaPairWeights"])


def test_end_to_end_fvmatrix_solve_oracle(oracle):
    """The reference's own fvScalarMatrix::solve (real faceAreaPairGAMGAgglomeration from libfiniteVolume,
    weights from mesh.Sf()) against glue restatement + C solver restatement: same V-cycle / iteration
    counts, residuals to 1e-6, solution to 1e-9."""
    g = load("fvsolve_box_14x12x10")
    p = solve_problem(g)
    S = oracle.System(p)
    x, perf = S.solve(p["psi"], p["source"], solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                      nCellsInCoarsestLevel=10, mergeLevels=1, tolerance=1e-10, relTol=0)
    r = g["ref_gamg_perf"]
    assert perf["nIterations"] == int(r[2]) and perf["converged"]
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-9 * np.max(np.abs(g["ref_gamg_psi"]))
    x, perf = S.solve(p["psi"], p["source"], solver="PCG", precond="DIC", tolerance=1e-10, relTol=0)
    r = g["ref_pcg_perf"]
    assert perf["nIterations"] == int(r[2])
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_pcg_psi"])) <= 1e-9 * np.max(np.abs(g["ref_pcg_psi"]))


def two_rank_problem(g):
    return n_rank_problem(g)


def junctions(g):
    """cyclic pair m = patches (2m, 2m+1) couples boxes pairs[m] = (lower, upper); a row of boxes: pair b = (b, b+1)"""
    nB = int(g["nBoxes"]) if "nBoxes" in g else 2
    if "pairs" in g:
        return [(int(a_), int(b_)) for a_, b_ in g["pairs"]]
    return [(b_, b_ + 1) for b_ in range(nB - 1)]


def n_rank_problem(g):
    """chain / grid fixture -> the rank-local problems of the equivalent N-rank run: rank r = cells
    [r*nHalf, (r+1)*nHalf); the cyclic pair j<m>a / j<m>b (patches 2m, 2m+1) becomes one processor patch on the lower box
    (towards the upper) and one on the upper box (towards the lower); per rank the patches are listed in ascending
    neighbour rank, as decomposePar writes them - several patches towards the same rank (split junctions) in the order
    of their pairs, which is the order the library pairs them in (k-th with k-th, ldu_comm.cpp paired_patch)."""
    nP, nC, nH = int(g["nPatches"][0]), int(g["nCells"]), int(g["nHalf"])
    nB = int(g["nBoxes"]) if "nBoxes" in g else 2
    P = [dict(faceCells=g["p%d_faceCells" % p], internalCoeffs=g["p%d_internalCoeffs" % p],
              boundaryCoeffs=g["p%d_boundaryCoeffs" % p], coupled=bool(g["p%d_coupled" % p][0]), pnf=None)
         for p in range(nP)]
    J = junctions(g)
    nJ = 2 * len(J)
    assert all(p["coupled"] for p in P[:nJ]) and not any(p["coupled"] for p in P[nJ:])
    diag = fv_oracle.add_boundary_diag(g["diag"], P)                      # fvScalarMatrix.C:152-153
    source = fv_oracle.add_boundary_source(g["source"], P, couples=False)  # :155-156
    l, u = g["lowerAddr"], g["upperAddr"]
    w = g["faceAreaPairWeights"]
    subs = []
    for r in range(nB):
        lo, hi = r * nH, (r + 1) * nH
        fsel = (l >= lo) & (l < hi)
        assert np.all((u[fsel] >= lo) & (u[fsel] < hi))
        mine = sorted([(b_ if a_ == r else a_, m, 2 * m + (0 if a_ == r else 1)) for m, (a_, b_) in enumerate(J) if r in (a_, b_)])
        patches = []
        for nbr, m, pi in mine:
            q = P[pi]
            assert np.all((q["faceCells"] >= lo) & (q["faceCells"] < hi))
            patches.append(dict(faceCells=(q["faceCells"] - lo).astype(np.int32), bouCoeffs=q["boundaryCoeffs"],
                                intCoeffs=q["internalCoeffs"], nbrDom=nbr, nbrRank=nbr, pair=m))
        sp = dict(nCells=hi - lo, lowerAddr=(l[fsel] - lo).astype(np.int32),
                  upperAddr=(u[fsel] - lo).astype(np.int32), diag=diag[lo:hi].copy(),
                  upper=g["upper"][fsel].copy(), source=source[lo:hi].copy(), psi=np.zeros(hi - lo),
                  faceWeights=w[fsel].copy(), patches=patches,
                  patches_dev=[dict(faceCells=q["faceCells"], nbrRank=q["nbrRank"]) for q in patches])
        if "lower" in g:
            sp["lower"] = g["lower"][fsel].copy()
        subs.append(sp)
    for r in range(nB):    # pairing for the oracle: my patch of pair m <-> the neighbour's patch of pair m
        for q in subs[r]["patches"]:
            nb = q["nbrDom"]
            q["nbrPatch"] = [j for j, q2 in enumerate(subs[nb]["patches"]) if q2["pair"] == q["pair"]][0]
    return subs


def cyclic_problem(g):
    """chain fixture as ONE domain whose coupled patches are the cyclic pairs themselves
    (cyclicLduInterface: patch 2b pairs with patch 2b+1)"""
    nP, nC = int(g["nPatches"][0]), int(g["nCells"])
    nB = int(g["nBoxes"]) if "nBoxes" in g else 2
    P = [dict(faceCells=g["p%d_faceCells" % p], internalCoeffs=g["p%d_internalCoeffs" % p],
              boundaryCoeffs=g["p%d_boundaryCoeffs" % p], coupled=bool(g["p%d_coupled" % p][0]), pnf=None)
         for p in range(nP)]
    nJ = 2 * len(junctions(g))
    diag = fv_oracle.add_boundary_diag(g["diag"], P)
    source = fv_oracle.add_boundary_source(g["source"], P, couples=False)
    patches = [dict(faceCells=P[p]["faceCells"].astype(np.int32), bouCoeffs=P[p]["boundaryCoeffs"],
                    intCoeffs=P[p]["internalCoeffs"], nbrDom=0, nbrRank=-1, nbrPatch=p ^ 1, cyclic=True)
               for p in range(nJ)]
    sp = dict(nCells=nC, lowerAddr=g["lowerAddr"], upperAddr=g["upperAddr"], diag=diag, upper=g["upper"],
              source=source, psi=np.zeros(nC), faceWeights=g["faceAreaPairWeights"], patches=patches,
              patches_dev=[dict(faceCells=q["faceCells"], nbrRank=-1, nbrPatch=q["nbrPatch"], cyclic=True)
                           for q in patches])
    if "lower" in g:
        sp["lower"] = g["lower"]
    return sp


CHAINS = ["fvsolve2_halves_6x8x7", "fvsolve4_chain_5x6x6", "fvsolve3_chain_asym_5x7x6",
          "fvsolve3_chain_nonblocking_4x7x6",
          # round 4: 3-D blocks (2 x 2 x 2: three coupled patches per rank) and two patches per pair of ranks
          "fvsolve8_blocks_2x2x2_4x4x4", "fvsolve2_split_halves_5x6x6", "fvsolve4_blocks_2x2x1_split_4x4x5"]


def SMOOTHER(name):
    return "nonBlockingGaussSeidel" if "nonblocking" in name else "GaussSeidel"


@pytest.mark.parametrize("name", CHAINS)
def test_two_rank_algorithm_against_reference_cyclic_emulation(oracle, name):
    """8(e) pin: the reference itself, in ONE process, solves N identical boxes in a row coupled only by
    cyclic patch pairs (real cyclicFvPatchField / cyclicGAMGInterface) - arithmetically an N-rank run with
    processor patches (2 ranks: one patch each; 4 ranks: the middle ranks have two).  The multi-domain oracle (rank-local DIC / GaussSeidel / agglomeration,
    interface updates, rank-ordered sums) must reproduce it: same V-cycle / iteration counts, residuals to
    1e-6, solution to 1e-8.  (Not bit-exact: the reference sums over all cells in one sequence, ranks sum
    locally and then add.)"""
    g = load(name)
    subs = n_rank_problem(g)
    S = oracle.System(subs)
    b = np.concatenate([s["source"] for s in subs])
    x, perf = S.solve(np.zeros(b.size), b, solver="GAMG", smoother=SMOOTHER(name), agglomerator="faceAreaPair",
                      nCellsInCoarsestLevel=10, mergeLevels=1, tolerance=1e-10, relTol=0)
    r = g["ref_gamg_perf"]
    assert perf["nIterations"] == int(r[2]) and perf["converged"]
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-8 * np.max(np.abs(g["ref_gamg_psi"]))
    kry = dict(solver="PBiCG", precond="DILU") if "asym" in name else dict(solver="PCG", precond="DIC")
    x, perf = S.solve(np.zeros(b.size), b, tolerance=1e-10, relTol=0, **kry)
    r = g["ref_pcg_perf"]
    assert perf["nIterations"] == int(r[2])
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_pcg_psi"])) <= 1e-8 * np.max(np.abs(g["ref_pcg_psi"]))


@pytest.mark.parametrize("name", CHAINS)
def test_cyclic_patches_single_domain_oracle(oracle, name):
    """cyclic coupled patches inside ONE domain (cyclicLduInterface / cyclicGAMGInterface): the same
    fixtures, solved as the reference solved them - one process, cyclic pairs.  nCellsInCoarsestLevel is
    the combined 10 x nBoxes the reference used."""
    g = load(name)
    sp = cyclic_problem(g)
    nB = int(g["nBoxes"]) if "nBoxes" in g else 2
    S = oracle.System([sp])
    x, perf = S.solve(sp["psi"], sp["source"], solver="GAMG", smoother=SMOOTHER(name), agglomerator="faceAreaPair",
                      nCellsInCoarsestLevel=10 * nB, mergeLevels=1, tolerance=1e-10, relTol=0)
    r = g["ref_gamg_perf"]
    assert perf["nIterations"] == int(r[2]) and perf["converged"]
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-9)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-11 * np.max(np.abs(g["ref_gamg_psi"]))
    kry = dict(solver="PBiCG", precond="DILU") if "asym" in name else dict(solver="PCG", precond="DIC")
    x, perf = S.solve(sp["psi"], sp["source"], tolerance=1e-10, relTol=0, **kry)
    r = g["ref_pcg_perf"]
    assert perf["nIterations"] == int(r[2])
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-9)


def test_rotational_cyclic_is_a_plain_cyclic_for_a_scalar_field(oracle):
    """round 6: a quarter annulus whose two ends are a ROTATIONAL cyclic pair (fv_case.box_mesh sector=True;
    `transform rotational` in the boundary file, forwardT() set by cyclicPolyPatch).  For a scalar field the interface multiplies the
    neighbour values by pow(diag(forwardT).component(cmpt), rank()) with rank() = 0 (cyclicLduInterfaceField.C:45-63): by 1.  The
    reference's own GAMG and PCG solves on that mesh are reproduced by the oracle with the pair as an ordinary cyclic pair - what
    the plug-in now hands the library for rank-0 fields (hipLduSolvers.C hipCheckTransforms)."""
    g = load("fvsolve_sector_8x6x5")
    assert bool(g["p0_coupled"][0]) and bool(g["p1_coupled"][0]) and not bool(g["p2_coupled"][0])
    sp = cyclic_problem(g)
    S = oracle.System([sp])
    x, perf = S.solve(sp["psi"], sp["source"], solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                      nCellsInCoarsestLevel=10, mergeLevels=1, tolerance=1e-10, relTol=0)
    r = g["ref_gamg_perf"]
    assert perf["nIterations"] == int(r[2]) and perf["converged"]
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-9)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-11 * np.max(np.abs(g["ref_gamg_psi"]))
    x, perf = S.solve(sp["psi"], sp["source"], solver="PCG", precond="DIC", tolerance=1e-10, relTol=0)
    r = g["ref_pcg_perf"]
    assert perf["nIterations"] == int(r[2])
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-9)
    for sm in ("GaussSeidel", "nonBlockingGaussSeidel"):
        assert np.array_equal(S.smooth(sm, g["smooth_x0"], sp["source"], 3), g["ref_smooth_" + sm]), sm


LU_CHAINS = ["fvsolve4_chain_lu_5x6x6", "fvsolve3_chain_asym_lu_5x7x6", "fvsolve8_blocks_lu_2x2x2_4x4x4"]


@pytest.mark.parametrize("name", LU_CHAINS)
def test_direct_solve_coarsest_with_coupled_patches(oracle, name):
    """directSolveCoarsest with coupled patches (round 6; GAMGSolver.C:95-106): the reference, in ONE process, factorises the
    coarsest level of N boxes coupled by cyclic pairs with its LUscalarMatrix (LUscalarMatrix.C:128-187) - 30 ... 80 cells.
    (a) the single-domain oracle with the cyclic patches reproduces it as closely as every other serial solve (1e-9);
    (b) the N-domain oracle - the gathered matrix of an N-rank run, LUscalarMatrix.C:190-318: the same dense matrix, cells in
        rank order - reproduces it to the multi-rank bar (rank-ordered sums elsewhere in the V-cycle)."""
    g = load(name)
    nB = int(g["nBoxes"]) if "nBoxes" in g else 2
    r = g["ref_gamg_perf"]
    sp = cyclic_problem(g)
    x, perf = oracle.System([sp]).solve(sp["psi"], sp["source"], solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                                        nCellsInCoarsestLevel=10 * nB, mergeLevels=1, tolerance=1e-10, relTol=0,
                                        directSolveCoarsest=1)
    assert perf["nIterations"] == int(r[2]) and perf["converged"]
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-9)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-11 * np.max(np.abs(g["ref_gamg_psi"]))
    subs = n_rank_problem(g)
    b = np.concatenate([s["source"] for s in subs])
    x, perf = oracle.System(subs).solve(np.zeros(b.size), b, solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                                        nCellsInCoarsestLevel=10, mergeLevels=1, tolerance=1e-10, relTol=0, directSolveCoarsest=1)
    assert perf["nIterations"] == int(r[2]) and perf["converged"]
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-8 * np.max(np.abs(g["ref_gamg_psi"]))
    # the LU changes the solve: the same fixture without it takes a different number of V-cycles or lands elsewhere
    x2, perf2 = oracle.System(subs).solve(np.zeros(b.size), b, solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                                          nCellsInCoarsestLevel=10, mergeLevels=1, tolerance=1e-10, relTol=0)
    assert perf2["nIterations"] != perf["nIterations"] or not np.array_equal(x, x2)


@pytest.mark.parametrize("name", CHAINS)
def test_smoothers_with_coupled_interfaces_bitexact(oracle, name):
    """GaussSeidel and nonBlockingGaussSeidel (3 sweeps) on the coupled systems, as run by the reference's
    own smoother classes with real cyclic interfaces: BIT-exact, both as one domain with cyclic patches and
    as N ranks with processor patches (every cell accumulates source / lower neighbours / interface terms
    in the same order in both settings)."""
    g = load(name)
    x0 = g["smooth_x0"]
    assert not np.array_equal(g["ref_smooth_GaussSeidel"], g["ref_smooth_nonBlockingGaussSeidel"]) \
        or "nonblocking" not in name
    for make in (cyclic_problem, n_rank_problem):
        sub = make(g)
        subs = [sub] if isinstance(sub, dict) else sub
        S = oracle.System(subs)
        b = np.concatenate([s["source"] for s in subs])
        for sm in ("GaussSeidel", "nonBlockingGaussSeidel"):
            assert np.array_equal(S.smooth(sm, x0, b, 3), g["ref_smooth_" + sm]), (make.__name__, sm)


def test_vector_fvmatrix_glue_oracle_matches_reference():
    """fvMatrix<vector> (U-equation like, fixedValue / zeroGradient / cyclic patches): addBoundaryDiag per
    component, addBoundarySource, A (cmptAv), the generic H, relax - bit-exact against the reference."""
    g = load("fvglueV_box_5x6x4_cyclic")
    P = glue_patches(g)
    l, u = g["lowerAddr"], g["upperAddr"]
    eq = np.array_equal
    assert any(p["coupled"] for p in P)
    for k in range(3):
        assert eq(fv_oracle.add_boundary_diag_cmpt(g["diag"], P, k), g["ref_addBoundaryDiag%d" % k])
    assert eq(fv_oracle.add_boundary_source_v(g["source"], P), g["ref_addBoundarySource"])
    assert eq(fv_oracle.add_boundary_source_v(g["source"], P, couples=False), g["ref_addBoundarySource_nocouples"])
    assert eq(fv_oracle.fvm_A_v(g["diag"], P, g["V"]), g["ref_A"])
    assert eq(fv_oracle.fvm_H_v(g["diag"], g["source"], l, u, g["upper"], g["lower"], g["psi"], P, g["V"]), g["ref_H"])
    d, s = fv_oracle.relax_v(0.7, g["diag"], g["source"], l, u, g["upper"], g["lower"], g["psi"], P)
    assert eq(d, g["ref_relax_diag"]) and eq(s, g["ref_relax_source"])


def coupled_problem(g):
    """What fvMatrix<vector>::solveCoupled hands to the LduMatrix<vector,scalar,scalar> solvers
    (fvMatrixSolve.C:236-249): diag + addBoundaryDiag(., 0), source + addBoundarySource(., false), and on the
    coupled (cyclic) interfaces the component-0 boundary / internal coefficients."""
    P = glue_patches(g)
    diag = fv_oracle.add_boundary_diag_cmpt(g["diag"], P, 0)
    source = fv_oracle.add_boundary_source_v(g["source"], P, couples=False)
    cp = [i for i, q in enumerate(P) if q["coupled"]]
    assert cp == [0, 1]
    patches = [dict(faceCells=P[i]["faceCells"].astype(np.int32), bouCoeffs=np.ascontiguousarray(P[i]["boundaryCoeffs"][:, 0]),
                    intCoeffs=np.ascontiguousarray(P[i]["internalCoeffs"][:, 0]), nbrDom=0, nbrRank=-1, nbrPatch=i ^ 1,
                    cyclic=True) for i in cp]
    return dict(nCells=int(g["nCells"]), lowerAddr=g["lowerAddr"], upperAddr=g["upperAddr"], diag=diag,
                upper=g["upper"], lower=g["lower"], source=source, psi=g["psi"], patches=patches,
                faceWeights=np.ones(g["lowerAddr"].size),
                patches_dev=[dict(faceCells=q["faceCells"], nbrRank=-1, nbrPatch=q["nbrPatch"], cyclic=True)
                             for q in patches])


COUPLED_KW = dict(preconditioner="DILU", tolerance=1e-9, relTol=0.0, maxIter=40, nSweeps=2)


def test_type_coupled_solve_oracle_matches_reference(oracle):
    """8f rank 4 end to end: the reference's own fvVectorMatrix::solve with `type coupled;` (PBiCCCG, PBiCICG,
    SmoothSolver on a convection-diffusion U equation with cyclic patches) against the coupled oracle fed by
    the fvMatrix glue oracle - bit for bit."""
    g = load("fvglueV_box_5x6x4_cyclic")
    sp = coupled_problem(g)
    S = oracle.System(sp)
    for solver in ("PBiCCCG", "PBiCICG", "SmoothSolver"):
        x, perf = S.c_solve(sp["psi"], sp["source"], solver=solver, **COUPLED_KW)
        assert np.array_equal(x, g["ref_coupled_" + solver].reshape(-1, 3)), solver
        assert perf["nIterations"] > 3


def test_segregated_vector_solve_oracle_matches_reference(oracle):
    """fvMatrix<vector>::solveSegregated (fvMatrixSolve.C:103-218) restated with the glue oracle: the source with
    every boundary contribution, per component the diagonal with that component's internal coefficients, the
    explicit coupled part taken back out through the interface update, then a scalar PBiCG/DILU solve with the
    component's interface coefficients - against the reference's own fvVectorMatrix::solve (cyclic patches)."""
    g = load("fvglueV_box_5x6x4_cyclic")
    P = glue_patches(g)
    nC = int(g["nCells"])
    src_all = fv_oracle.add_boundary_source_v(g["source"], P, couples=True)
    cp = [i for i, q in enumerate(P) if q["coupled"]]
    x = np.zeros((nC, 3))
    for c in range(3):
        diag = fv_oracle.add_boundary_diag_cmpt(g["diag"], P, c)
        src = src_all[:, c].copy()
        psi = g["psi"][:, c].copy()
        for i in cp:   # initMatrixInterfaces / updateMatrixInterfaces(bouCoeffsCmpt, ..., sourceCmpt)
            nb = P[i ^ 1]
            for f, cell in enumerate(P[i]["faceCells"]):
                src[cell] -= P[i]["boundaryCoeffs"][f, c] * psi[nb["faceCells"][f]]
        patches = [dict(faceCells=P[i]["faceCells"].astype(np.int32), bouCoeffs=np.ascontiguousarray(P[i]["boundaryCoeffs"][:, c]),
                        intCoeffs=np.ascontiguousarray(P[i]["internalCoeffs"][:, c]), nbrDom=0, nbrPatch=i ^ 1) for i in cp]
        sp = dict(nCells=nC, lowerAddr=g["lowerAddr"], upperAddr=g["upperAddr"], diag=diag, upper=g["upper"],
                  lower=g["lower"], patches=patches)
        xc, perf = oracle.System(sp).solve(psi, src, solver="PBiCG", precond="DILU", tolerance=1e-10, relTol=0, maxIter=60)
        assert perf["converged"]
        x[:, c] = xc
    ref = g["ref_segregated_PBiCG"].reshape(-1, 3)
    assert np.abs(x - ref).max() <= 1e-12 * np.abs(ref).max()


def stencil_patches(g):
    return [dict(faceCells=g["ref_p%d_faceCells" % p], value=g["ref_p%d_value" % p], Cf=g["ref_p%d_Cf" % p])
            for p in range(int(g["ref_nPatches"][0]))]


@pytest.mark.parametrize("name", NAMES)
def test_higher_order_schemes_oracle_matches_reference(name):
    """8f rank 2: linearUpwind<scalar>::correction and cellLimited Gauss linear (k = 1, 0.5) gradients
    against the reference's own scheme classes - bit-exact."""
    g = load(name)
    l, u = g["lowerAddr"], g["upperAddr"]
    C, Cf, g0 = g["ref_C"], g["ref_Cf"], g["ref_gaussLinearGrad"]
    assert np.array_equal(g0, g["ref_gaussGrad"])     # zero boundary values: the basic gradient of the fixture
    assert np.array_equal(fv_oracle.linear_upwind_correction(l, u, g["phi"], C, Cf, g0), g["ref_linearUpwind_correction"])
    P = stencil_patches(g)
    lim1 = fv_oracle.cell_limited_grad(1.0, l, u, g["vf"], C, Cf, g0, P)
    assert np.array_equal(lim1, g["ref_cellLimitedGrad_k1"])
    assert not np.array_equal(lim1, g0)               # the limiter is active somewhere
    assert np.array_equal(fv_oracle.cell_limited_grad(0.5, l, u, g["vf"], C, Cf, g0, P), g["ref_cellLimitedGrad_k05"])
    # vector forms: linearUpwindV<vector>::correction and cellLimitedGrad<vector> on U with non-zero patch values
    gU = g["ref_gaussLinearGradU"]
    assert np.array_equal(fv_oracle.linear_upwind_v_correction(l, u, g["phi"], g["ref_weights"], g["U"], C, Cf, gU),
                          g["ref_linearUpwindV_correction"])
    PU = [dict(faceCells=g["ref_p%d_faceCells" % p].astype(int), value=g["ref_p%d_valueU" % p], Cf=g["ref_p%d_Cf" % p])
          for p in range(int(g["ref_nPatches"][0]))]
    limV = fv_oracle.cell_limited_grad_v(1.0, l, u, g["U"], C, Cf, gU, PU)
    assert np.array_equal(limV, g["ref_cellLimitedGradV_k1"]) and not np.array_equal(limV, gU)
    assert np.array_equal(fv_oracle.cell_limited_grad_v(0.5, l, u, g["U"], C, Cf, gU, PU), g["ref_cellLimitedGradV_k05"])
    # the `bounded` wrapper
    PP = [dict(faceCells=g["ref_p%d_faceCells" % p].astype(int), phi=g["ref_p%d_phi" % p])
          for p in range(int(g["ref_nPatches"][0]))]
    assert np.array_equal(fv_oracle.bounded_sp(g["ref_div_upwind_diag_bphi"], l, u, g["phi"], PP, g["ref_V"]),
                          g["ref_div_bounded_upwind_diag"])
