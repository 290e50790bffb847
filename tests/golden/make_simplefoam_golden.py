"""Generates tests/golden/simplefoam_pitzdaily*.json: every `Solving for` line of the REFERENCE's own simpleFoam
(oracle/_ref/simpleFoam = applications/solvers/incompressible/simpleFoam/simpleFoam.C linked against the reference's
libfiniteVolume / turbulence / transport / fvOptions units and libOpenFOAM by oracle/build_ref_fv.sh; no plug-in) on
BASELINE config C2, pitzDaily (oracle/pitzdaily_case.py), 40 SIMPLE iterations - with the tutorial's PCG + DIC for p and
with the motorBike GAMG block (BASELINE: "GAMG p-solve") - and tests/golden/simplefoam_motorbike_tut.json: the same binary on
the tutorial-size motorBike mesh of the reference's own blockMesh + snappyHexMesh (oracle/motorbike_simplefoam_case.py; ~321 k
cells - the fixture records which mesh it was made on, snappyHexMesh is not reproducible across hosts -, 12 SIMPLE iterations, the tutorial's GAMG block for p, smoothSolver + GaussSeidel for U / k / epsilon).  Run here (needs /root/reference for the build); JSON = data.
  python tests/golden/make_simplefoam_golden.py"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import cavity_case as cc
import pitzdaily_case as pc
import motorbike_simplefoam_case as mc

STEPS = 40
MB_STEPS = 12

if __name__ == "__main__":
    if not pc.available():
        raise SystemExit("oracle/_ref/simpleFoam missing: run oracle/build_ref_fv.sh (needs /root/reference)")
    for tag, psolver in (("", None), ("_gamg", pc.GAMG)):
        with tempfile.TemporaryDirectory() as d:
            case = os.path.join(d, "pitzDaily")
            pc.write(case, STEPS, p_solver=psolver)
            lines = cc.solve_lines(pc.run(case))
        out = dict(case="simpleFoam pitzDaily 12 225 cells, %d SIMPLE iterations, kEpsilon%s" % (STEPS, ", p: GAMG GaussSeidel faceAreaPair" if psolver else ", p: PCG DIC"),
                   generator="tests/golden/make_simplefoam_golden.py", lines=lines)
        json.dump(out, open(os.path.join(HERE, "simplefoam_pitzdaily%s.json" % tag), "w"))
        print(tag or "pcg", len(lines), lines[2], lines[-3])
    if mc.available():
        with tempfile.TemporaryDirectory() as d:
            case = os.path.join(d, "motorBike")
            mc.write(case, MB_STEPS)
            lines = cc.solve_lines(mc.run(case))
        out = dict(case="simpleFoam on the tutorial-size motorBike mesh (snappyHexMesh, castellated, %d cells), %d SIMPLE iterations, "
                        "kEpsilon, p: GAMG GaussSeidel faceAreaPair, U/k/epsilon: smoothSolver GaussSeidel" % (mc.mesh_identity()["nCells"], MB_STEPS),
                   generator="tests/golden/make_simplefoam_golden.py", mesh=mc.mesh_identity(), lines=lines)
        json.dump(out, open(os.path.join(HERE, "simplefoam_motorbike_tut.json"), "w"))
        print("motorBike", len(lines), lines[3], lines[-3])
    else:
        print("data/motorbike/mbtut_polymesh.npz missing (tools/make_motorbike.py mbtut --small): motorBike fixture not written")
