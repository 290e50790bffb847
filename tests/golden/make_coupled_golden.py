"""Generates tests/golden/coupled_*.npz from the REAL reference (oracle/_ref/ref_driver modes "cops" / "csolve":
the reference's own LduMatrix<vector,scalar,scalar> Amul / Tmul / residual / preconditioners / smoother / solvers).
Run in the container that has /root/reference built under oracle/_ref:  python tests/golden/make_coupled_golden.py
The fixtures hold inputs and the reference's outputs only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import conftest  # noqa: F401  (loads the package alias)
import oracle_py as O
from openfoam_amd import cases

CASES = {
    "coupled_box_asym_9x7x6": cases.box3d(9, 7, 6, asym=True),
    "coupled_rand_asym_700": cases.random_graph(700, asym=True),
    "coupled_box_sym_8x6x5": cases.box3d(8, 6, 5),
}
TOL = np.array([1e-8, 1e-7, 1e-9])
MAXITER = 70


def main():
    for name, p in CASES.items():
        rng = np.random.RandomState(17)
        n = p["nCells"]
        p["psiV"] = rng.randn(n * 3)
        p["sourceV"] = rng.randn(n * 3)
        out = {"p_" + k: np.asarray(v) for k, v in p.items() if k in
               ("nCells", "lowerAddr", "upperAddr", "diag", "upper", "lower", "faceWeights")}
        out["psiV"], out["sourceV"] = p["psiV"], p["sourceV"]
        out["tolerance"], out["maxIter"] = TOL, np.int32(MAXITER)
        ref, _ = O.run_ref("cops", p)
        out.update(ref)
        asym = "lower" in p
        combos = ([("PBiCCCG", "DILU"), ("PBiCICG", "DILU"), ("PBiCCCG", "diagonal"), ("SmoothSolver", "none")]
                  if asym else [("PCICG", "diagonal"), ("PCICG", "none"), ("SmoothSolver", "none")])
        for solver, pre in combos:
            d = ("solver %s; preconditioner %s; smoother GaussSeidel; tolerance (%r %r %r); relTol (0 0 0); "
                 "maxIter %d; nSweeps 2;" % (solver, pre, float(TOL[0]), float(TOL[1]), float(TOL[2]), MAXITER))
            r, _ = O.run_ref("csolve", p, d)
            out["solve_%s_%s_psi" % (solver, pre)] = r["psiV"]
            out["solve_%s_%s_perf" % (solver, pre)] = r["perf"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, sorted(out))


if __name__ == "__main__":
    main()
