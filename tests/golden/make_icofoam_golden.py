"""Generates tests/golden/icofoam_cavity40*.json: every `Solving for` line of the REFERENCE's own icoFoam
(oracle/_ref/icoFoam = applications/solvers/incompressible/icoFoam/icoFoam.C linked against the reference's libfiniteVolume
units and libOpenFOAM by oracle/build_ref_fv.sh; no plugin) on BASELINE config C1, the 40x40 lid-driven cavity
(oracle/cavity_case.py), 100 time steps.  Run here (needs /root/reference for the build); the JSON is data.
  python tests/golden/make_icofoam_golden.py"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import cavity_case as cc

GAMG = ("        solver          GAMG;\n        tolerance       1e-06;\n        relTol          0;\n"
        "        smoother        GaussSeidel;\n        nPreSweeps      0;\n        nPostSweeps     2;\n"
        "        cacheAgglomeration on;\n        agglomerator    faceAreaPair;\n        nCellsInCoarsestLevel 10;\n"
        "        mergeLevels     1;")

if __name__ == "__main__":
    if not cc.available():
        raise SystemExit("oracle/_ref/icoFoam missing: run oracle/build_ref_fv.sh (needs /root/reference)")
    for tag, psolver in (("", None), ("_gamg", GAMG)):
        with tempfile.TemporaryDirectory() as d:
            case = os.path.join(d, "cavity")
            cc.write(case, 40, 100, p_solver=psolver)
            lines = cc.solve_lines(cc.run(case))
        out = dict(case="icoFoam cavity 40x40x1, 100 steps, deltaT 0.0025" + (", p: GAMG GaussSeidel faceAreaPair" if psolver else ""),
                   generator="tests/golden/make_icofoam_golden.py", lines=lines)
        json.dump(out, open(os.path.join(HERE, "icofoam_cavity40%s.json" % tag), "w"))
        print(tag or "pcg", len(lines), lines[2], lines[-1])
